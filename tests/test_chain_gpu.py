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


@pytest.mark.parametrize("local", [False, True])
def test_level_chain_is_bitwise_the_per_launch_plan(local, monkeypatch):
    """cfg B (BASELINE.json configs[1]): 40 steps through the replayed graphs (five 8-step launches), once with the 4x4 / 2x2
    levels as persistent chains and once with the same plan taken apart again.  Samples, x0 predictions and the noise of the
    last step must be equal bit for bit - a stale byte anywhere in a hand-off between two stages of a chain would show here -
    and no chain wait may have timed out.
    local = False (LFVDM_CHAIN_LOCAL=0): the split-K tile body in every stage; taken apart = ``disable_chains``, one
    lfvdm_conv_igemm / lfvdm_gn_apply launch per stage with the same tune codes.
    local = True (default): sample-local stages; taken apart = ``split_chains``, every stage a chain of its own (the same
    body and work items, ordered by launch boundaries instead of flags).  The per-launch TILE plan sums K in another order:
    it must agree within the forward tolerance, not bitwise."""
    monkeypatch.setenv("LFVDM_CHAIN_LOCAL", "1" if local else "0")
    from improved_diffusion import _native as nat
    cfg, sd, inp = load_case("cfgB")
    model = build_native(cfg, sd)
    d = {k: v.cuda() for k, v in inp.items()}
    mk = dict(frame_indices=d["frame_indices"], obs_mask=d["obs_mask"], latent_mask=d["latent_mask"], x0=d["x0"])
    shape = tuple(inp["x"].shape)
    outs, launches = [], []
    for mode in ("chained", "apart") + (("tiles",) if local else ()):
        s = _sampler(model, shape)
        s.begin(d["x"].clone(), mk)
        assert getattr(s.plan, "tuned", False)
        assert len(s.plan.chains) >= 2, "the encoder and decoder halves of the low-resolution levels"
        kinds = [k for c in s.plan.chains for k in c["kinds"]]
        assert (nat.CHAIN_LOCAL in kinds) == local
        if mode != "chained":
            if local and mode == "apart":
                s.plan.split_chains()
            else:
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
        if mode == "chained":
            print("[chain] stages per chain:", [c["n"] for c in s.plan.chains], "kinds:", [c["kinds"] for c in s.plan.chains],
                  "work items:", [c["items"] for c in s.plan.chains], "grids:", [c["grid"] for c in s.plan.chains])
        del s
    print(f"[chain] launches per step: {launches[0]} chained, {launches[1]} per launch")
    assert launches[0] <= launches[1] - 15
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][0]).all())
    if local:       # 40 free-running steps against the tile plan: rounding only (same noise stream: bitwise equal draws)
        assert torch.equal(outs[0][2], outs[2][2])
        dev = float((outs[0][0] - outs[2][0]).abs().max())
        print(f"[chain] sample-local chains vs per-launch tile plan after 40 steps: max |d| = {dev:.3g}")
        assert dev <= 1e-4          # (observed 2.4e-7; a wrong GroupNorm half in one decoder stage gave ~1e-3 here)


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
    assert pl.split_chains() >= 20        # every stage a launch of its own: the same bodies, ordered by launch boundaries
    pl.launch()
    torch.cuda.synchronize()
    assert torch.equal(ref, pl.out)
    tile = Plan(model.native_engine(), B, T, H, W, False)       # and the plan without chains: rounding only
    tile.refresh_weights()
    tile.set_inputs(d["x"], d["x0"], torch.tensor([500.0, 20.0], device="cuda"), d["frame_indices"], d["obs_mask"], d["latent_mask"])
    tile.launch()
    torch.cuda.synchronize()
    assert float((tile.out - ref).abs().max()) <= 2e-4 + 1e-3 * float(ref.abs().max())


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


# ------------------------------------------------------------------------------------------- sample-local stages (round 6)
def _launch_chain(stage_list, timeout_s=2.0):
    """Plan and launch a chain of ChainStage objects once (tensors the stages point at are the caller's)."""
    import ctypes as C
    from improved_diffusion import _native as nat
    L = nat.lib()
    n = len(stage_list)
    stages = (nat.ChainStage * n)(*stage_list)
    cap = 1 << 18
    deps = (C.c_int32 * cap)()
    used, ws_f, cnt_i = C.c_int64(), C.c_int64(), C.c_int64()
    n_flags, grid, lds = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.lfvdm_chain_plan(stages, n, deps, cap, C.byref(used), C.byref(n_flags), C.byref(ws_f), C.byref(cnt_i), C.byref(grid),
                            C.byref(lds))
    assert rc == 0, rc
    assert grid.value <= L.lfvdm_chain_capacity(lds.value)
    ws = torch.empty(max(1, ws_f.value), device="cuda")
    cnt = torch.zeros(max(1, cnt_i.value), dtype=torch.int32, device="cuda")
    for i in range(n):
        cv = stages[i].conv
        cv.splitk_ws, cv.splitk_cnt = ws.data_ptr() + 4 * stages[i].ws_off, cnt.data_ptr() + 4 * stages[i].cnt_off
        cv.splitk_ws_floats, cv.splitk_cnt_ints = ws.numel() - stages[i].ws_off, cnt.numel() - stages[i].cnt_off
    stages_dev = torch.frombuffer(bytearray(bytes(memoryview(stages))), dtype=torch.uint8).cuda()
    deps_dev = torch.frombuffer(bytearray(bytes(memoryview(deps))[:4 * max(1, used.value)]), dtype=torch.int32).cuda()
    flags = torch.zeros(max(1, n_flags.value), dtype=torch.int32, device="cuda")
    ctl = torch.zeros(nat.CHAIN_CTL_INTS, dtype=torch.int32, device="cuda")
    for _ in range(2):          # twice: the generation counter advances, nothing is reset
        nat.check(L.lfvdm_level_chain(stages_dev.data_ptr(), n, deps_dev.data_ptr(), flags.data_ptr(), ctl.data_ptr(), grid.value,
                                      lds.value, timeout_s, nat.stream()), "lfvdm_level_chain")
    torch.cuda.synchronize()
    assert int(ctl[nat.CHAIN_CTL_ABORT].item()) == 0
    return [s.n_items for s in stages], grid.value


LOCAL_CASES = {
    # name: (N, Hs, Ho, ksize, stride, up, C0, C1, s2C0, s2C1, Cout, res, gn (None | (film, skip_raw, gw, ld)), rt)
    "res2x2_gn_film": (40, 2, 2, 3, 1, 0, 128, 0, 0, 0, 128, False, (True, 1, 0, 0), 1),
    "res2x2_residual_raw": (40, 2, 2, 3, 1, 0, 128, 0, 0, 0, 128, True, None, 1),
    "res2x2_residual_gn_both": (40, 2, 2, 3, 1, 0, 128, 0, 0, 0, 128, True, (False, 0, 0, 0), 1),
    "down8to4": (40, 8, 4, 3, 2, 0, 128, 0, 0, 0, 128, False, (False, 0, 0, 0), 2),
    "down8to4_rt1": (40, 8, 4, 3, 2, 0, 128, 0, 0, 0, 128, False, (False, 0, 0, 0), 1),
    "down4to2": (40, 4, 2, 3, 2, 0, 128, 0, 0, 0, 128, False, (False, 0, 0, 0), 1),
    "up2to4_cat_half": (40, 2, 4, 3, 1, 1, 128, 0, 0, 0, 128, False, (False, 0, 8, 256), 2),
    "skip_segment_cat_half": (40, 2, 2, 3, 1, 0, 128, 0, 128, 128, 128, False, (False, 0, 8, 256), 1),
    "skip_segment_4x4_rt2": (40, 4, 4, 3, 1, 0, 128, 0, 128, 64, 128, False, None, 2),
    "proj1x1_residual_gn": (40, 2, 2, 1, 1, 0, 128, 0, 0, 0, 128, True, (False, 0, 0, 0), 1),
    "res4x4_rt1": (40, 4, 4, 3, 1, 0, 128, 0, 0, 0, 128, False, (True, 1, 0, 0), 1),
    "res4x4_rt2": (40, 4, 4, 3, 1, 0, 128, 0, 0, 0, 128, True, (True, 0, 0, 0), 2),
    "batch1_ragged_rows": (6, 2, 2, 3, 1, 0, 128, 0, 0, 0, 128, True, (True, 0, 0, 0), 1),
    "ragged_rt2": (5, 4, 4, 3, 1, 0, 64, 0, 0, 0, 64, True, (True, 0, 0, 0), 2),
    "wide_256_gw8": (20, 2, 2, 3, 1, 0, 64, 0, 0, 0, 256, False, (True, 1, 0, 0), 1),
    "narrow_64_gw2": (20, 2, 2, 3, 1, 0, 64, 0, 0, 0, 64, True, (True, 0, 0, 0), 1),
    "wide_512_gw16_concat_source": (8, 2, 2, 3, 1, 0, 64, 64, 0, 0, 512, False, (False, 0, 0, 0), 1),
    "map4x1": (12, 2, 2, 1, 1, 0, 192, 0, 0, 0, 64, False, None, 1),
    "cin192_3x3_runtime_loop": (8, 2, 2, 3, 1, 0, 192, 0, 0, 0, 64, True, (True, 0, 0, 0), 1),      # 3 units per tap and wave: no instance
    "cin256_1x1": (12, 4, 4, 1, 1, 0, 256, 0, 0, 0, 128, True, (False, 0, 0, 0), 2),
    "cin512_1x1_runtime_loop": (6, 2, 2, 1, 1, 0, 256, 256, 0, 0, 64, False, None, 1),
    "cin64_3x3_rt2": (10, 4, 4, 3, 1, 0, 64, 0, 64, 0, 64, False, (True, 1, 0, 0), 2),
}


@pytest.mark.parametrize("name", sorted(LOCAL_CASES))
def test_sample_local_stage_matches_the_tile_kernel(name):
    """One LFVDM_CHAIN_LOCAL stage (csrc/conv_local_body.h: whole samples x 16 filters x all of K per work item, MFMA
    16x16x4, GroupNorm by lane butterflies) against the stand-alone implicit-GEMM launch on the same lfvdm_conv_args: every
    operand form the low-resolution levels use (unet.py:194-207 ResBlocks with FiLM, :91-114 / :60-88 resampling, the 1x1
    skip segment on a raw concat, residuals, the concat-half normalisation gn_gw / gn_ld, ragged last items).  fp32,
    different K order: |d| <= 2e-5 * (1 + |ref|max)."""
    import ctypes as C
    from improved_diffusion import _native as nat
    N, Hs, Ho, k, stride, up, C0, C1, s2C0, s2C1, Cout, res, gn, rt = LOCAL_CASES[name]
    T = 2 if N % 2 == 0 else 1
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()      # noqa: E731
    Cin, C2, taps = C0 + C1, s2C0 + s2C1, k * k
    M, Min = N * Ho * Ho, N * Hs * Hs
    src0, src1 = rn(Min, C0), (rn(Min, C1) if C1 else None)
    W, b = rn(Cout, taps, Cin) / (taps * Cin) ** 0.5, 0.1 * rn(Cout)
    s2a, s2b = (rn(M, s2C0) if s2C0 else None), (rn(M, s2C1) if s2C1 else None)
    W2, b2 = (rn(Cout, C2) / C2 ** 0.5 if C2 else None), (0.1 * rn(Cout) if C2 else None)
    resid = rn(M, Cout) if res else None
    gamma, beta, film = 1.0 + 0.1 * rn(Cout), 0.1 * rn(Cout), 0.3 * rn(N // T, 2 * Cout)
    outs = []
    for local in (False, True):
        a = nat.ConvArgs()
        a.src0, a.src1, a.C0, a.C1, a.N, a.Hs, a.Ws, a.Ho, a.Wo = src0.data_ptr(), src1.data_ptr() if C1 else None, C0, C1, N, Hs, Hs, Ho, Ho
        if name == "map4x1":
            a.Hs, a.Ws, a.Ho, a.Wo = 4, 1, 4, 1
        a.up, a.stride, a.ksize, a.W, a.bias, a.Cout = up, stride, k, W.data_ptr(), b.data_ptr(), Cout
        if C2:
            a.s2src0, a.s2src1, a.s2C0, a.s2C1 = s2a.data_ptr(), s2b.data_ptr() if s2C1 else None, s2C0, s2C1
            a.W2, a.bias2 = W2.data_ptr(), b2.data_ptr()
        if res:
            a.res, a.ldr = resid.data_ptr(), Cout
        raw = torch.full((M, Cout), 7.0, device="cuda")
        a.out, a.ldo, a.out_mode = raw.data_ptr(), Cout, nat.OUT_ROWS
        gout = None
        if gn:
            use_film, skip_raw, gw, ld = gn
            gout = torch.full((M, ld or Cout), 7.0, device="cuda")
            a.gn_gamma, a.gn_beta, a.gn_out = gamma.data_ptr(), beta.data_ptr(), gout.data_ptr()
            if use_film:
                a.gn_film, a.gn_film_ld, a.gn_film_div = film.data_ptr(), 2 * Cout, T
            else:
                a.gn_film_div = 1
            a.gn_act, a.gn_skip_raw, a.gn_eps, a.gn_gw, a.gn_ld = nat.ACT_SILU, skip_raw, 1e-5, gw, ld
        if local:
            assert nat.lib().lfvdm_chain_local_ok(C.byref(a), rt) == 0
            st = nat.ChainStage()
            st.kind, st.cfg = nat.CHAIN_LOCAL, rt
            C.memmove(C.byref(st.conv), C.byref(a), C.sizeof(nat.ConvArgs))
            items, grid = _launch_chain([st])
            assert items[0] == -(-M // (16 * rt)) * (Cout // 16) and grid % 8 == 0
        else:
            ws, cnt = nat.splitk_workspace(torch.device("cuda", torch.cuda.current_device()))
            a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
            a.tune = 0
            nat.check(nat.lib().lfvdm_conv_igemm(C.byref(a), nat.stream()), "lfvdm_conv_igemm")
            torch.cuda.synchronize()
        outs.append((raw, gout))
    (r0, g0), (r1, g1) = outs
    if not (gn and gn[1]):
        tol = 2e-5 * (1.0 + float(r0.abs().max()))
        assert float((r0 - r1).abs().max()) <= tol, (name, float((r0 - r1).abs().max()))
    else:
        assert bool((r1 == 7.0).all()), "gn_skip_raw: the raw tensor is not written"
    if gn:
        tol = 2e-5 * (1.0 + float(g0.abs().max()))
        assert float((g0 - g1).abs().max()) <= tol, (name, float((g0 - g1).abs().max()))
        if gn[3]:
            assert bool((g1[:, Cout:] == 7.0).all()), "the other half of the concat operand is someone else's"
