"""Persistent level chain (csrc/level_chain.hip) on the MI355X: ONE launch for a run of low-resolution stages must give
BITWISE the per-launch plan (same tile codes: every stage keeps its K-slice order and epilogue), replay after replay."""
import os

import pytest
import torch

from test_oracle_golden import load_case
from test_forward_gpu import build_native
from test_sampler_gpu import make_diffusion

pytestmark = pytest.mark.gpu


def _sampler(model, shape, steps="", inject=False):
    from improved_diffusion.gaussian_diffusion import GraphSampler
    return GraphSampler(make_diffusion(1000, steps), model, shape, True, inject_noise=inject)


def test_level_chain_is_bitwise_the_per_launch_plan():
    """cfg B (BASELINE.json configs[1]): 40 steps through the replayed graphs (five 8-step launches), once with the 4x4 / 2x2
    levels as persistent chains and once with the same plan taken apart again (``disable_chains``: one launch per stage, the
    same tune codes).  Samples, x0 predictions and the noise of the last step must be equal bit for bit - a stale byte
    anywhere in a hand-off between two stages of a chain would show here - and no chain wait may have timed out."""
    cfg, sd, inp = load_case("cfgB")
    model = build_native(cfg, sd)
    d = {k: v.cuda() for k, v in inp.items()}
    mk = dict(frame_indices=d["frame_indices"], obs_mask=d["obs_mask"], latent_mask=d["latent_mask"], x0=d["x0"])
    shape = tuple(inp["x"].shape)
    outs, launches = [], []
    for chained in (True, False):
        s = _sampler(model, shape)
        s.begin(d["x"].clone(), mk)
        assert getattr(s.plan, "tuned", False)
        assert len(s.plan.chains) >= 2, "the encoder and decoder halves of the low-resolution levels"
        if not chained:
            s.plan.disable_chains()
            s.graph = s.graph_k = None
            s.begin(d["x"].clone(), mk)
            assert not s.plan.chains
        s.seed.fill_(4242)
        out = s.run(999, 40)
        torch.cuda.synchronize()
        assert not s.plan.chains_aborted()
        outs.append((out["sample"].clone(), out["pred_xstart"].clone(), s.noise.clone()))
        launches.append(len(s.plan.steps) + s.extra_launches)
        if chained:
            print("[chain] stages per chain:", [c["n"] for c in s.plan.chains], "work items:", [c["items"] for c in s.plan.chains],
                  "grids:", [c["grid"] for c in s.plan.chains])
        del s
    print(f"[chain] launches per step: {launches[0]} chained, {launches[1]} per launch")
    assert launches[0] <= launches[1] - 15
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][0]).all())


def test_level_chain_eager_launches_repeat_bitwise():
    """The chain outside a graph: 30 eager forward launches on the same inputs (generation counter advancing every launch,
    flags never reset) give the same output every time, equal to the per-launch plan's."""
    cfg, sd, inp = load_case("cfgB")
    model = build_native(cfg, sd)
    d = {k: v.cuda() for k, v in inp.items()}
    mk = dict(frame_indices=d["frame_indices"], obs_mask=d["obs_mask"], latent_mask=d["latent_mask"], x0=d["x0"])
    from improved_diffusion._engine import Plan
    B, T, _, H, W = inp["x"].shape
    pl = Plan(model.native_engine(), B, T, H, W, False)
    pl.refresh_weights()
    pl.set_inputs(d["x"], d["x0"], torch.tensor([500.0, 20.0], device="cuda"), d["frame_indices"], d["obs_mask"], d["latent_mask"])
    pl.launch()
    pl.autotune()
    assert pl.chains
    ref = None
    for i in range(30):
        pl.launch()
        torch.cuda.synchronize()
        o = pl.out.clone()
        if ref is None:
            ref = o
        assert torch.equal(ref, o), i
    epoch = [int(c["ctl"][0].item()) for c in pl.chains]
    assert min(epoch) >= 30
    pl.disable_chains()
    pl.launch()
    torch.cuda.synchronize()
    assert torch.equal(ref, pl.out)


def test_a_raised_abort_word_makes_the_sampler_fall_back(capfd):
    """Every wait inside a chain is bounded: a poller that times out raises the chain's abort word, every other poller sees
    it and leaves.  The host side of that contract: ``p_sample_loop`` finds the word raised after the chain, says so on
    stderr, takes the plan apart into one launch per stage and runs the chain again - the caller gets valid samples."""
    from improved_diffusion import _native as nat
    cfg, sd, inp = load_case("cfgB")
    model = build_native(cfg, sd)
    d = {k: v.cuda() for k, v in inp.items()}
    mk = dict(frame_indices=d["frame_indices"], obs_mask=d["obs_mask"], latent_mask=d["latent_mask"], x0=d["x0"])
    shape = tuple(inp["x"].shape)
    diff = make_diffusion(1000, "25")
    torch.manual_seed(11)
    a, _ = diff.p_sample_loop(model, shape, clip_denoised=True, model_kwargs=mk, return_decoded=False)
    s = diff._graph_sampler(model, shape, True)
    assert s.plan.chains and not s.chain_timed_out()
    s.plan.chains[0]["ctl"][nat.CHAIN_CTL_ABORT] = 1            # what a timed-out poller does
    b, _ = diff.p_sample_loop(model, shape, clip_denoised=True, model_kwargs=mk, return_decoded=False)
    err = capfd.readouterr().err
    assert "persistent level chain timed out" in err
    assert not s.plan.chains and s.chain_timeouts == 1 and bool(torch.isfinite(b).all())
    n_steps = len(s.plan.steps)
    c, _ = diff.p_sample_loop(model, shape, clip_denoised=True, model_kwargs=mk, return_decoded=False)      # stays per launch
    assert len(s.plan.steps) == n_steps and not s.plan.chains and bool(torch.isfinite(c).all())
