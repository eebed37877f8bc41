"""Training path on the MI355X: gradients of the native U-Net vs the CPU oracle's autograd (which is
pinned to the reference's gradients by tests/golden/backward_*.npz).  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe, unet_oracle as uo
from conftest import GOLDEN
from test_oracle_golden import load_case
from test_forward_gpu import build_native

pytestmark = pytest.mark.gpu


_oracle_grads = {}


def oracle_gradients(name, cfg, sd, inp, probe):
    """CPU autograd of the oracle (minutes for the 119 M-parameter pixel model: computed once per case)."""
    if name not in _oracle_grads:
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        o, _ = uo.unet_forward(sdo, cfg, inp["x"], inp["x0"], inp["t"].float(), inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])
        (o * probe).sum().backward()
        _oracle_grads[name] = {k: v.grad for k, v in sdo.items()}
    return _oracle_grads[name]


@pytest.mark.parametrize("inplace", [False, True], ids=["autograd", "inplace"])
@pytest.mark.parametrize("name", ["micro", "micro_rb2", "micro_px", "cfgC", "cfgE_T2"])
def test_parameter_gradients_match_oracle(name, inplace):
    """Both gradient delivery modes (_backward._GradMode): returned to autograd (default), or accumulated into p.grad
    by the kernels (what TrainLoop switches on).  cfgC is the BASELINE.json configs[2] training shape (ch128, 4 levels,
    20 frames of which 3 are padding, batch 2): there the in-place mode also takes the grouped RPE path.  cfgE_T2 is the
    pixel-space model of configs[4] (the reference's published training recipe, README.md:54-57: 128x128x3, ch128, rb2,
    5 levels, head dims 96 / 128, 119 M parameters) on 2 frames: large-map GroupNorm backward, weight gradients over
    M = 32768 rows, head Cout = 3.  micro_px: the same 3-channel head on a small model."""
    g = np.load(os.path.join(GOLDEN, f"backward_{name}.npz"))
    cfg, sd, inp = load_case(name)
    model = build_native(cfg, sd).train()
    model.native_grad_accumulation = inplace
    d = {k: v.cuda() for k, v in inp.items()}
    probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
    out, _ = model(d["x"], x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"],
                   obs_mask=d["obs_mask"], latent_mask=d["latent_mask"])
    # the differentiable path must agree with the no-grad engine
    with torch.no_grad():
        out_ng, _ = model(d["x"], x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"],
                          obs_mask=d["obs_mask"], latent_mask=d["latent_mask"])
    assert torch.allclose(out.detach(), out_ng, atol=2e-4), float((out.detach() - out_ng).abs().max())
    (out * probe.cuda()).sum().backward()
    if name == "cfgC" and inplace:
        from improved_diffusion import _backward as bw
        assert bw._rpe_group.state is not None and bw._embed.state is not None, "grouped embedding / RPE path expected"
    # oracle gradients (CPU autograd)
    ograds = oracle_gradients(name, cfg, sd, inp, probe)
    gmax = float(g["gmax"])
    keys = [str(k) for k in g["keys"]]
    worst, worst_key = 0.0, None
    for i, (k, p) in enumerate(model.named_parameters()):
        assert k == keys[i]
        ref = ograds[k]
        assert p.grad is not None, k
        err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-3 * gmax)
        if err > worst:
            worst, worst_key = err, k
        # and against the REAL reference's per-tensor gradient norms
        assert abs(float(p.grad.double().norm()) - float(g["norms"][i])) < 3e-3 * (float(g["norms"][i]) + 1e-3 * gmax * np.sqrt(ref.numel())), k
    print(f"[{name}] worst relative gradient error vs oracle: {worst:.2e} ({worst_key})")
    assert worst < 2e-3, (worst, worst_key)
    # gradient w.r.t. the input latents against the reference's
    if not inplace:
        xg = d["x"].clone().requires_grad_(True)
        model.zero_grad(set_to_none=True)
        o2, _ = model(xg, x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"], obs_mask=d["obs_mask"],
                      latent_mask=d["latent_mask"])
        (o2 * probe.cuda()).sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), g["dx"], atol=5e-4 * np.abs(g["dx"]).max())


def test_dropout_training_matches_oracle_with_the_same_draws():
    """dropout > 0 (unet.py:166): the native training path draws its own keep masks; with those masks handed to the
    oracle, outputs and gradients must agree; eval mode ignores dropout."""
    from improved_diffusion import _backward as bw
    name = "micro"
    cfg, sd, inp = load_case(name)
    model = build_native(dict(cfg, dropout=0.25), sd).train()
    d = {k: v.cuda() for k, v in inp.items()}
    kw = dict(x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"], obs_mask=d["obs_mask"],
              latent_mask=d["latent_mask"])
    probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
    bw.dropout_log = []
    try:
        torch.manual_seed(5)
        out, _ = model(d["x"], **kw)
        keeps = list(bw.dropout_log)
        with torch.no_grad():                       # train() mode without gradients still drops
            bw.dropout_log = []
            torch.manual_seed(5)
            out_ng, _ = model(d["x"], **kw)
            assert len(bw.dropout_log) == len(keeps) > 0
    finally:
        bw.dropout_log = None
    assert torch.equal(out.detach(), out_ng)
    frac = float(torch.cat([(k == 0).float().flatten() for k in keeps]).mean())
    assert 0.2 < frac < 0.3, frac
    assert all(bool(((k == 0) | ((k - 1 / 0.75).abs() < 1e-6)).all()) for k in keeps)
    (out * probe.cuda()).sum().backward()
    # oracle with the same draws: rows (n, y, x) x C  ->  (N, C, H, W)
    B, T, _, H, W = inp["x"].shape
    keeps_cpu, res = [], []
    for k in keeps:
        C = k.shape[1]
        hw = k.shape[0] // (B * T)
        side = int(round(hw ** 0.5))
        keeps_cpu.append(k.cpu().view(B * T, side, side, C).permute(0, 3, 1, 2).contiguous())
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o, _ = uo.unet_forward(sdo, cfg, inp["x"], inp["x0"], inp["t"].float(), inp["frame_indices"], inp["obs_mask"],
                           inp["latent_mask"], dropout_keep=keeps_cpu)
    assert torch.allclose(out.detach().cpu(), o.detach(), atol=2e-4, rtol=1e-3), float((out.detach().cpu() - o).abs().max())
    (o * probe).sum().backward()
    gmax = max(float(v.grad.abs().max()) for v in sdo.values())
    worst = 0.0
    for k, p in model.named_parameters():
        ref = sdo[k].grad
        worst = max(worst, float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-3 * gmax))
    assert worst < 2e-3, worst
    # eval(): dropout is the identity
    model.eval()
    with torch.no_grad():
        e1, _ = model(d["x"], **kw)
    ref_eval, _ = uo.unet_forward(sd, cfg, inp["x"], inp["x0"], inp["t"].float(), inp["frame_indices"], inp["obs_mask"],
                                  inp["latent_mask"])
    assert torch.allclose(e1.cpu(), ref_eval, atol=2e-4, rtol=1e-3)


def test_distributed_data_parallel_wrap_fires_hooks_and_matches_plain_gradients():
    """the model must be wrappable by DistributedDataParallel (reference train_util.py:116-125).  One
    rank is enough to exercise the reducer: it needs every parameter's AccumulateGrad hook to fire on every step."""
    from improved_diffusion import dist_util
    dist_util.setup_dist()
    cfg, sd, inp = load_case("micro")
    d = {k: v.cuda() for k, v in inp.items()}
    kw = dict(x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"], obs_mask=d["obs_mask"],
              latent_mask=d["latent_mask"])
    probe = torch.from_numpy(recipe.gaussianish("micro/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32)).cuda()
    plain = build_native(cfg, sd).train()
    out, _ = plain(d["x"], **kw)
    (out * probe).sum().backward()
    model = build_native(cfg, sd).train()
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, broadcast_buffers=False,
                                                    bucket_cap_mb=128, find_unused_parameters=False)
    fired = {}
    for k, p in model.named_parameters():
        p.register_hook(lambda g, k=k: fired.__setitem__(k, fired.get(k, 0) + 1))
    for _ in range(2):              # the second step fails in the reducer if a hook was missed in the first
        ddp.zero_grad(set_to_none=True)
        out, _ = ddp(d["x"], **kw)
        (out * probe).sum().backward()
    assert all(fired.get(k, 0) == 2 for k, _ in model.named_parameters()), [k for k, _ in model.named_parameters() if fired.get(k, 0) != 2][:5]
    gmax = max(float(q.grad.abs().max()) for q in plain.parameters())
    for (k, p), (_, q) in zip(model.named_parameters(), plain.named_parameters()):
        # wgrad partial tiles are added with float atomics: run-to-run differences of a few ulp of the largest terms
        # (biases in front of a GroupNorm have gradients that cancel to ~0: floor by the global gradient scale)
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-4 * float(q.grad.abs().max()) + 1e-4 * gmax), k


def test_grouped_rpe_training_path_matches_per_network_path(monkeypatch):
    """T*T >= 32: the RPE networks of a training step run as grouped launches (lfvdm_rpe_nets with stored activations,
    lfvdm_rpe_nets_bwd, lfvdm_conv_wgrad_grouped).  Every parameter gradient must agree with the per-network path
    (itself pinned to the oracle above), and with the oracle's gradients for the RPE parameters."""
    cfg, sd, _ = load_case("micro")
    B, T, C, H = 2, 8, cfg["in_channels"], 16
    g = torch.Generator().manual_seed(11)
    x, x0 = torch.randn(B, T, C, H, H, generator=g), torch.randn(B, T, C, H, H, generator=g)
    t = torch.tensor([37.0, 512.0])
    fi = torch.stack([torch.sort(torch.randperm(60, generator=g)[:T]).values for _ in range(B)])
    obs = torch.zeros(B, T, 1, 1, 1); obs[:, :3] = 1
    lat = 1 - obs; lat[:, -1] = 0
    probe = torch.randn(B, T, cfg["out_channels"], H, H, generator=g)
    d = dict(x=x.cuda(), x0=x0.cuda(), t=t.cuda(), fi=fi.cuda(), obs=obs.cuda(), lat=lat.cuda())
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LFVDM_RPE_GROUPED", mode)
        model = build_native(cfg, sd).train()
        model.native_grad_accumulation = True
        out, _ = model(d["x"], x0=d["x0"], timesteps=d["t"], frame_indices=d["fi"], obs_mask=d["obs"], latent_mask=d["lat"])
        (out * probe.cuda()).sum().backward()
        grads[mode] = {k: p.grad.clone() for k, p in model.named_parameters()}
        if mode == "1":
            from improved_diffusion import _backward as bw
            assert bw._rpe_group.state is not None, "the grouped path should have been taken"
    gmax = max(float(v.abs().max()) for v in grads["0"].values())
    for k in grads["0"]:
        a, b = grads["1"][k], grads["0"][k]
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()) + 1e-4 * gmax), (k, float((a - b).abs().max()))
    # RPE parameters against the oracle's autograd as well
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o, _ = uo.unet_forward(sdo, cfg, x, x0, t, fi, obs, lat)
    (o * probe).sum().backward()
    for k in grads["1"]:
        if ".rpe_" in k:
            ref = sdo[k].grad
            err = float((grads["1"][k].cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-3 * gmax)
            assert err < 2e-3, (k, err)


def test_attention_maps_in_grad_mode_match_no_grad_engine():
    """return_attn_weights=True while gradients are recorded (reference unet.py:428-464 returns the detached
    |mean over heads| maps in every mode): same maps as the no-grad engine, and the backward still works."""
    cfg, sd, inp = load_case("micro")
    model = build_native(cfg, sd).train()
    d = {k: v.cuda() for k, v in inp.items()}
    kw = dict(x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"], obs_mask=d["obs_mask"],
              latent_mask=d["latent_mask"])
    out, attn = model(d["x"], return_attn_weights=True, **kw)
    with torch.no_grad():
        out_ng, attn_ng = model(d["x"], return_attn_weights=True, **kw)
    assert set(attn) == {"spatial", "temporal", "mixed"} and attn["mixed"] == []
    assert len(attn["temporal"]) == len(attn_ng["temporal"]) > 0 and len(attn["spatial"]) == len(attn_ng["spatial"]) > 0
    for kind in ("temporal", "spatial"):
        for a, b in zip(attn[kind], attn_ng[kind]):
            assert a.shape == b.shape and not a.requires_grad
            assert torch.allclose(a, b, atol=1e-4), (kind, float((a - b).abs().max()))     # other tile shapes than the tuned plan
    out.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


@pytest.mark.parametrize("deterministic", ["0", "1"], ids=["atomics", "deterministic"])
def test_three_channel_head_gradients_in_both_reduction_modes(deterministic, monkeypatch):
    """Head with Cout = 3 (pixel space): the narrow weight-gradient kernel's bias epilogue handles four columns per lane
    and must guard each on its own - with LFVDM_DETERMINISTIC=1 an unguarded fourth column lands in the NEXT slice's slab
    row (a wrong, racing bias gradient), with atomics one float past the gradient arena.  In-place delivery (what
    TrainLoop uses), gradients vs the oracle, and the float after the last gradient must stay untouched."""
    monkeypatch.setenv("LFVDM_DETERMINISTIC", deterministic)
    name = "micro_px"
    g = np.load(os.path.join(GOLDEN, f"backward_{name}.npz"))
    cfg, sd, inp = load_case(name)
    model = build_native(cfg, sd).train()
    model.native_grad_accumulation = True
    # gradients in ONE flat buffer with a canary behind the last one (out.2.bias is the last parameter)
    params = list(model.parameters())
    flat = torch.zeros(sum(p.numel() for p in params) + 64, device="cuda")
    flat[-64:] = 12345.0
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    d = {k: v.cuda() for k, v in inp.items()}
    probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
    runs = []
    for _ in range(2):
        flat[:-64].zero_()
        out, _ = model(d["x"], x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"],
                       obs_mask=d["obs_mask"], latent_mask=d["latent_mask"])
        (out * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        runs.append(flat.clone())
    assert bool((flat[-64:] == 12345.0).all()), "write past the last gradient"
    if deterministic == "1":
        assert torch.equal(runs[0], runs[1]), "deterministic mode must be bitwise reproducible"
    ograds = oracle_gradients(name, cfg, sd, inp, probe)
    gmax = float(g["gmax"])
    for k, p in model.named_parameters():
        ref = ograds[k]
        err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-3 * gmax)
        assert err < 2e-3, (k, err)
    ob = dict(model.named_parameters())["out.2.bias"].grad.cpu()
    np.testing.assert_allclose(ob.numpy(), ograds["out.2.bias"].numpy(), rtol=2e-4, atol=2e-4 * float(ograds["out.2.bias"].abs().max()))


def test_skip_slot_handoff_equals_autograd_sums_and_detects_undrained_slots(monkeypatch):
    """The skip-connection gradient hand-off (_backward._SkipSlot) against plain autograd sums (LFVDM_NO_SKIP_SLOTS), and
    its safety net: a filled slot that nobody drained by the end of the backward pass raises instead of dropping the
    decoder's contribution."""
    from improved_diffusion import _backward as bw
    cfg, sd, inp = load_case("micro")
    d = {k: v.cuda() for k, v in inp.items()}
    kw = dict(x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"], obs_mask=d["obs_mask"],
              latent_mask=d["latent_mask"])
    probe = torch.from_numpy(recipe.gaussianish("micro/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32)).cuda()
    grads = {}
    for slots in (True, False):
        monkeypatch.setattr(bw, "_SKIP_SLOTS", slots)
        model = build_native(cfg, sd).train()
        model.native_grad_accumulation = True
        out, _ = model(d["x"], **kw)
        (out * probe).sum().backward()
        torch.cuda.synchronize()
        assert bw._SkipSlot.pending == [] and not bw._SkipSlot.queued
        grads[slots] = {k: p.grad.clone() for k, p in model.named_parameters()}
    gmax = max(float(v.abs().max()) for v in grads[False].values())
    for k in grads[False]:
        a, b = grads[True][k], grads[False][k]
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max()) + 1e-4 * gmax), (k, float((a - b).abs().max()))
    # a slot left full at the end of a backward pass is an error, and it is cleared for the next pass
    slot = bw._SkipSlot()
    leaf = torch.ones(4, device="cuda", requires_grad=True)

    class Fill(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, gr):
            slot.put(gr.clone())
            return gr * 2
    with pytest.raises(RuntimeError, match="never ran"):
        Fill.apply(leaf).sum().backward()
    assert slot.g is None and bw._SkipSlot.pending == []
