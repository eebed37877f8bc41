"""Generate tests/golden/*.npz by IMPORTING the real reference (build container only).

    python oracle/make_golden.py            # needs /root/reference, never runs on the GPU box

TEST INFRASTRUCTURE.  Nothing of the reference's source is copied: this script only imports
its modules, feeds them the closed-form parameters/inputs of oracle/recipe.py and stores the
numeric outputs.  It also checks oracle/{unet,diffusion}_oracle.py against the reference on
the spot and prints the deviations (the same check runs from the fixtures in
tests/test_oracle_golden.py).  diffusion_space="pixel" is used so that no VAE is fetched
(SURVEY fact 5); arithmetic is identical for pre-encoded latents.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from improved_diffusion import gaussian_diffusion as rgd  # noqa: E402  (reference)
from improved_diffusion import respace as rrespace  # noqa: E402
from improved_diffusion import rpe as rrpe  # noqa: E402
from improved_diffusion import nn as rnn  # noqa: E402
from improved_diffusion import unet as runet  # noqa: E402
from improved_diffusion import script_util as rsu  # noqa: E402

from oracle import recipe, unet_oracle as uo, diffusion_oracle as do  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)

CONFIGS = {
    # name: (cfg kwargs, B, T, H, n_pad)
    "micro": (dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)), 2, 4, 16, 1),
    "micro_rb2": (dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(2,), num_res_blocks=2,
                       num_heads=2), 1, 3, 8, 0),
    "cfgA": (dict(model_channels=32, channel_mult=(1, 2, 2, 2), attention_resolutions=(2, 4)), 1, 5, 32, 0),
    # pixel-like micro model: 3 input / 3 output channels (the head's Cout is not a multiple of 4)
    "micro_px": (dict(in_channels=3, model_channels=64, channel_mult=(1, 2), attention_resolutions=(2,)), 1, 3, 16, 0),
    "cfgB": (dict(model_channels=64, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 20, 16, 3),
    "cfgB_T14": (dict(model_channels=64, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 14, 16, 0),
    # BASELINE.json configs[2]: the per-GPU training workload (ch128, 4 levels, 20 frames of which 3 are padding, batch 2)
    "cfgC": (dict(model_channels=128, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 20, 16, 3),
    # BASELINE.json configs[4] (pixel space 128x128x3, num_channels=128, reference defaults: num_res_blocks=2,
    # channel_mult (1,1,2,3,4), attention at 16x16 and 8x8 -> head dims 96 and 128) on 2 frames
    "cfgE_T2": (dict(in_channels=3, model_channels=128, num_res_blocks=2, channel_mult=(1, 1, 2, 3, 4),
                     attention_resolutions=(8, 16)), 1, 2, 128, 0),
}


def build_reference_model(cfg):
    m = runet.UNetVideoModel(
        in_channels=cfg["in_channels"], model_channels=cfg["model_channels"], out_channels=cfg["out_channels"],
        num_res_blocks=cfg["num_res_blocks"], attention_resolutions=cfg["attention_resolutions"],
        dropout=0.0, channel_mult=cfg["channel_mult"], num_heads=cfg["num_heads"],
        use_scale_shift_norm=cfg["use_scale_shift_norm"], use_rpe_net=cfg["use_rpe_net"])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ours = uo.param_shapes(cfg)
    assert list(shapes.keys()) == list(ours.keys()), "state-dict key order mismatch"
    assert shapes == ours, "state-dict shape mismatch"
    assert [n for n, _ in m.named_parameters()] == list(ours.keys())
    sd = recipe.fill_state_dict(shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    return m, {k: torch.from_numpy(v) for k, v in sd.items()}


def tt(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def maxdiff(a, b):
    return float((a - b).abs().max())


def gen_forward(only=None):
    for name, (kw, B, T, H, n_pad) in CONFIGS.items():
        if only and name not in only:
            continue
        cfg = uo.make_cfg(**kw)
        model, sd = build_reference_model(cfg)
        inp = recipe.make_inputs(name, B, T, cfg["in_channels"], H, H, n_pad=n_pad)
        ti = tt(inp)
        ts = ti["t"].float() * 1.0  # already "rescaled" timesteps 0..999
        with torch.no_grad():
            ref, attn = model(ti["x"], x0=ti["x0"], timesteps=ts, frame_indices=ti["frame_indices"],
                              obs_mask=ti["obs_mask"], latent_mask=ti["latent_mask"], return_attn_weights=True)
            mine, attn2 = uo.unet_forward(sd, cfg, ti["x"], ti["x0"], ts, ti["frame_indices"],
                                          ti["obs_mask"], ti["latent_mask"], return_attn_weights=True)
        d = maxdiff(ref, mine)
        da = max(maxdiff(a, b) for k in ("spatial", "temporal") for a, b in zip(attn[k], attn2[k]))
        print(f"[forward {name}] out rms {float(ref.pow(2).mean().sqrt()):.4f}  oracle-vs-ref max|d| {d:.3e}  attn {da:.3e}")
        # fp32 re-association noise only: the reference itself sits 4e-5..1.4e-4 from an fp64 evaluation
        assert d < 1e-4 and da < 5e-5
        np.savez_compressed(
            os.path.join(OUT, f"forward_{name}.npz"), out=ref.numpy(),
            attn_t0=attn["temporal"][0].numpy()[:8], attn_s0=attn["spatial"][0].numpy()[:1, :32, :32],
            x_sum=np.float64(inp["x"].astype(np.float64).sum()), frame_indices=inp["frame_indices"],
            n_params=np.int64(sum(v.numel() for v in sd.values())))


def gen_backward(names=("micro", "micro_rb2", "cfgC")):
    """loss = sum(out * probe); parameter gradients of reference vs oracle; compact per-tensor
    summaries are stored (full tensors would be MBs)."""
    for name in names:
        kw, B, T, H, n_pad = CONFIGS[name]
        cfg = uo.make_cfg(**kw)
        model, sd = build_reference_model(cfg)
        inp = tt(recipe.make_inputs(name, B, T, cfg["in_channels"], H, H, n_pad=n_pad))
        probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
        ts = inp["t"].float()
        x = inp["x"].clone().requires_grad_(True)
        out, _ = model(x, x0=inp["x0"], timesteps=ts, frame_indices=inp["frame_indices"],
                       obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"])
        (out * probe).sum().backward()
        ref_g = {k: p.grad.clone() for k, p in model.named_parameters()}
        sd2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        x2 = inp["x"].clone().requires_grad_(True)
        out2, _ = uo.unet_forward(sd2, cfg, x2, inp["x0"], ts, inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])
        (out2 * probe).sum().backward()
        # Some gradients are analytically zero (a per-channel bias in front of a GroupNorm whose
        # groups are single channels at ch=32): they come out as pure rounding noise, so the
        # relative error is floored by 1e-3 of the largest gradient entry of the whole model.
        gmax = max(float(g.abs().max()) for g in ref_g.values())
        worst = 0.0
        for k in ref_g:
            scale = float(ref_g[k].abs().max()) + 1e-3 * gmax
            worst = max(worst, maxdiff(ref_g[k], sd2[k].grad) / scale)
        print(f"[backward {name}] worst relative grad diff oracle-vs-ref {worst:.3e}; dx {maxdiff(x.grad, x2.grad):.3e}")
        assert worst < 1e-3
        keys = list(ref_g.keys())
        np.savez_compressed(
            os.path.join(OUT, f"backward_{name}.npz"),
            keys=np.array(keys), norms=np.array([float(ref_g[k].double().norm()) for k in keys]),
            sums=np.array([float(ref_g[k].double().sum()) for k in keys]),
            head=np.stack([np.resize(ref_g[k].flatten()[:8].numpy(), 8) for k in keys]),
            gmax=np.float64(gmax), dx=x.grad.numpy())


def gen_ops():
    """Op-level fixtures from the reference's own modules."""
    out = {}
    # timestep embedding (nn.py:105-123)
    t = torch.tensor([0.0, 1.0, 37.5, 999.0])
    out["temb_t"] = t.numpy()
    out["temb_64"] = rnn.timestep_embedding(t, 64).numpy()
    assert maxdiff(rnn.timestep_embedding(t, 64), uo.timestep_embedding(t, 64)) == 0.0
    # RPE.forward_safe_qk scalar definition (rpe.py:85-96) vs the einsum (rpe.py:72-74)
    rp = rrpe.RPE(channels=32, num_heads=2, time_embed_dim=128, use_rpe_net=True)
    shapes = {k: tuple(v.shape) for k, v in rp.state_dict().items()}
    rp.load_state_dict(tt(recipe.fill_state_dict({("ops/rpe." + k): s for k, s in shapes.items()}))
                       if False else {k: torch.from_numpy(recipe.fill_param("ops/rpe." + k, s)) for k, s in shapes.items()})
    B, D, Hh, T, Fh = 2, 3, 2, 5, 16
    qk = torch.from_numpy(recipe.gaussianish("ops/rpe/qk", B * D * Hh * T * Fh).reshape(B, D, Hh, T, Fh).astype(np.float32))
    temb = torch.from_numpy(recipe.gaussianish("ops/rpe/temb", B * T * 128).reshape(B * T, 128).astype(np.float32))
    fi = torch.tensor([[0, 1, 2, 3, 4], [3, 17, 18, 400, 999]])
    rel = fi.unsqueeze(-1) - fi.unsqueeze(-2)
    with torch.no_grad():
        safe = rp.forward_safe_qk(qk, rel, temb)
        fast = rp.forward_qk(qk, rel, temb)
        R = rp.get_R(rel, temb)
    assert maxdiff(safe, fast) < 1e-5
    sdr = {"p.rpe_net." + k.split("rpe_net.")[1]: v for k, v in rp.state_dict().items()}
    assert maxdiff(R, uo.rpe_net(sdr, "p.rpe_net", temb, rel, 2)) < 1e-6
    out["rpe_safe_qk"] = safe.numpy()
    out["rpe_R"] = R.numpy()
    out["rpe_fi"] = fi.numpy()
    # one ResBlock with Cin != Cout (unet.py:117-207)
    rb = runet.ResBlock(64, 128, 0.0, out_channels=32, use_scale_shift_norm=True)
    shapes = {k: tuple(v.shape) for k, v in rb.state_dict().items()}
    rsd = {k: torch.from_numpy(recipe.fill_param("ops/res." + k, s)) for k, s in shapes.items()}
    rb.load_state_dict(rsd)
    x = torch.from_numpy(recipe.gaussianish("ops/res/x", 3 * 64 * 8 * 8).reshape(3, 64, 8, 8).astype(np.float32))
    emb = torch.from_numpy(recipe.gaussianish("ops/res/emb", 3 * 128).reshape(3, 128).astype(np.float32))
    with torch.no_grad():
        y = rb(x, emb)
    assert maxdiff(y, uo.res_block({"p." + k: v for k, v in rsd.items()}, "p", x, emb)) < 1e-5
    out["res_y"] = y.numpy()
    # temporal + spatial attention instances (rpe.py:99-174)
    for kind in ("temporal", "spatial"):
        use_rpe = kind == "temporal"
        att = rrpe.RPEAttention(channels=64, num_heads=4, time_embed_dim=128, use_rpe_net=True,
                                use_rpe_q=use_rpe, use_rpe_k=use_rpe, use_rpe_v=use_rpe)
        shapes = {k: tuple(v.shape) for k, v in att.state_dict().items()}
        asd = {k: torch.from_numpy(recipe.fill_param(f"ops/att_{kind}." + k, s)) for k, s in shapes.items()}
        att.load_state_dict(asd)
        B, D, C, T = 2, 6, 64, 5
        xa = torch.from_numpy(recipe.gaussianish(f"ops/att_{kind}/x", B * D * C * T).reshape(B, D, C, T).astype(np.float32))
        mask = torch.tensor([[1., 1, 1, 0, 0], [1, 0, 1, 1, 0]]) if use_rpe else None
        # the time embedding as the network produces it: one row per batch element, repeated over its T frames
        # (unet.py:440 expands the timesteps over T) - `att_temb` holds these rows
        temb_att = temb.view(B, T, -1)[:, 0].repeat_interleave(T, 0)
        with torch.no_grad():
            ya, aa = att._forward(xa, temb_att, fi if use_rpe else None, mask)
            yo, ao = uo.rpe_attention({"p." + k: v for k, v in asd.items()}, "p", xa, temb_att,
                                      fi if use_rpe else None, mask, 4, use_rpe)
        assert maxdiff(ya, yo) < 1e-5 and maxdiff(aa, ao) < 1e-6
        out[f"att_{kind}_y"] = ya.numpy()
        out[f"att_{kind}_attn"] = aa.numpy()
    out["att_temb"] = temb_att.numpy()
    # Down / Up (unet.py:60-114)
    for kind, mod in (("down", runet.Downsample(32, True)), ("up", runet.Upsample(32, True))):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        msd = {k: torch.from_numpy(recipe.fill_param(f"ops/{kind}." + k, s)) for k, s in shapes.items()}
        mod.load_state_dict(msd)
        xu = torch.from_numpy(recipe.gaussianish(f"ops/{kind}/x", 2 * 32 * 8 * 8).reshape(2, 32, 8, 8).astype(np.float32))
        with torch.no_grad():
            out[f"{kind}_y"] = mod(xu).numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("[ops] ok")


def gen_diffusion():
    """Tables, q_sample, training_losses, p_mean_variance and a 5-step p_sample trajectory
    (SURVEY §8c item 4).  The noise p_sample draws internally is the recorded recipe noise
    (randn_like is swapped for the duration of the call, in this script only)."""
    out = {}
    pixel = {"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None}
    table_names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                   "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                   "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                   "posterior_mean_coef2"]
    for tag, steps, resp in (("lin1000", 1000, ""), ("lin32", 32, ""), ("lin1000_r250", 1000, "250"),
                             ("cos100_r10_15", 100, "10,15")):
        sched = "cosine" if tag.startswith("cos") else "linear"
        diff = rsu.create_gaussian_diffusion(steps=steps, noise_schedule=sched, timestep_respacing=resp,
                                             rescale_timesteps=True, rescale_learned_sigmas=True,
                                             diffusion_space_kwargs=dict(pixel))
        base = do.cosine_betas(steps) if sched == "cosine" else do.linear_betas(steps)
        tab = do.Tables(base, do.space_timesteps(steps, resp) if resp else None)
        for n in table_names:
            ref = getattr(diff, n)
            assert np.array_equal(ref, getattr(tab, n)), (tag, n)
            out[f"{tag}/{n}"] = ref
        assert diff.timestep_map == tab.timestep_map
        out[f"{tag}/timestep_map"] = np.array(diff.timestep_map)
    # model-driven fixtures on the micro model
    name = "micro"
    kw, B, T, H, n_pad = CONFIGS[name]
    cfg = uo.make_cfg(**kw)
    model, sd = build_reference_model(cfg)
    inp = tt(recipe.make_inputs(name, B, T, cfg["in_channels"], H, H, n_pad=n_pad))
    shape = inp["x"].shape
    noise = [torch.from_numpy(recipe.gaussianish(f"diff/noise{i}", inp["x"].numel()).reshape(shape).astype(np.float32))
             for i in range(6)]
    mk = dict(frame_indices=inp["frame_indices"], obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"], x0=inp["x0"])
    for tag, steps, resp in (("lin1000", 1000, ""), ("lin1000_r250", 1000, "250")):
        diff = rsu.create_gaussian_diffusion(steps=steps, timestep_respacing=resp, rescale_timesteps=True,
                                             rescale_learned_sigmas=True, diffusion_space_kwargs=dict(pixel))
        tab = do.Tables(do.linear_betas(steps), do.space_timesteps(steps, resp) if resp else None)
        nt = diff.num_timesteps
        t = torch.tensor([nt - 1, nt // 3])[:B]
        with torch.no_grad():
            xq = diff.q_sample(inp["x0"], t, noise=noise[0])
            losses = diff.training_losses(model, inp["x0"], t, model_kwargs=mk, noise=noise[0],
                                          latent_mask=1 - inp["obs_mask"], eval_mask=inp["latent_mask"])
            pmv = diff.p_mean_variance(model, inp["x"], t, clip_denoised=True, model_kwargs=mk)
        assert maxdiff(xq, do.q_sample(tab, inp["x0"], t, noise[0])) < 1e-6
        out[f"{tag}/t"] = t.numpy()
        out[f"{tag}/q_sample"] = xq.numpy()
        for k, v in losses.items():
            out[f"{tag}/loss/{k}"] = v.numpy()

        def model_fn(x_t, ts):
            return uo.unet_forward(sd, cfg, x_t, inp["x0"], ts, inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])[0]

        with torch.no_grad():
            mine = do.training_losses(tab, model_fn, inp["x0"], t, noise[0], 1 - inp["obs_mask"], inp["latent_mask"])
            eps = model_fn(inp["x"], do.model_timesteps(tab, t))
            pm = do.p_mean_variance(tab, eps, inp["x"], t)
        for k in losses:
            assert maxdiff(losses[k], mine[k]) < 1e-5, k
        for k in ("mean", "variance", "log_variance", "pred_xstart"):
            # x0-hat = sqrt(1/acp)*x - sqrt(1/acp - 1)*eps amplifies the eps rounding noise by up to
            # 157x at t=999 before the clamp (gaussian_diffusion.py:341-346)
            amp = 1.0 + float(tab.sqrt_recipm1_alphas_cumprod[t.numpy()].max()) if k == "pred_xstart" else 1.0
            assert maxdiff(pmv[k], pm[k]) < 5e-5 * amp, (k, maxdiff(pmv[k], pm[k]))
            out[f"{tag}/pmv/{k}"] = pmv[k].numpy() if k in ("mean", "pred_xstart") else pmv[k].numpy()[:, :1, :1, :1, :1]
        # 5-step trajectory from the top of the chain
        real_randn_like = torch.randn_like
        traj, x_ref, x_mine = [], inp["x"].clone(), inp["x"].clone()
        for j, i in enumerate(range(nt - 1, nt - 6, -1)):
            ti = torch.tensor([i] * B)
            torch.randn_like = lambda x, _n=noise[j + 1]: _n
            try:
                with torch.no_grad():
                    x_ref = diff.p_sample(model, x_ref, ti, clip_denoised=True, model_kwargs=mk)["sample"]
            finally:
                torch.randn_like = real_randn_like
            with torch.no_grad():
                x_mine, _ = do.p_sample(tab, model_fn(x_mine, do.model_timesteps(tab, ti)), x_mine, ti, noise[j + 1])
            traj.append(x_ref.numpy())
            assert maxdiff(x_ref, x_mine) < 5e-5 * (j + 1), (tag, j, maxdiff(x_ref, x_mine))
        # last step of the chain (t == 0: no noise added)
        ti = torch.zeros(B, dtype=torch.long)
        torch.randn_like = lambda x, _n=noise[0]: _n
        try:
            with torch.no_grad():
                x_last = diff.p_sample(model, inp["x"], ti, clip_denoised=True, model_kwargs=mk)["sample"]
        finally:
            torch.randn_like = real_randn_like
        out[f"{tag}/traj"] = np.stack(traj)
        out[f"{tag}/p_sample_t0"] = x_last.numpy()
        print(f"[diffusion {tag}] ok")
    np.savez_compressed(os.path.join(OUT, "diffusion.npz"), **out)


def gen_decode():
    """The encode / decode boundary (reference gaussian_diffusion.py:914-947) with a stand-in autoencoder
    (oracle/fake_vae.py) and a synthetic stats dict: de-normalisation of pre-encoded latents, chunked decode, chunked
    encode.  The reference object is built in pixel space (its 'latent' constructor fetches the SVD pipeline by name,
    which must never run) and then switched to the latent attributes its encode/decode read; ``Tensor.cuda`` is the
    identity for the duration of the calls (this container has no GPU; the reference moves chunks with ``.cuda()``)."""
    from oracle import fake_vae
    diff = rsu.create_gaussian_diffusion(steps=1000, diffusion_space_kwargs={
        "diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None})
    st = fake_vae.stats_dict(4)
    diff.diffusion_space = "latent"
    diff.pre_encoded = True
    diff.pre_encoded_stats_dict = {"mean": st["mean"].reshape(1, 1, -1, 1, 1), "std": st["std"].reshape(1, 1, -1, 1, 1)}
    diff.vae, diff.image_processor = fake_vae.FakeVAE(), fake_vae.FakeImageProcessor()
    diff.enc_dec_dtype, diff.original_dtype = torch.float32, torch.float32
    z = torch.from_numpy(recipe.gaussianish("decode/z", 2 * 7 * 4 * 4 * 4).reshape(2, 7, 4, 4, 4).astype(np.float32))
    px = torch.from_numpy((0.5 * recipe.gaussianish("decode/px", 2 * 5 * 3 * 16 * 16)).reshape(2, 5, 3, 16, 16).astype(np.float32)).clamp(-1, 1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        dec_pre = diff.decode(z, chunk_size=4)                 # de-normalise, then 14 frames in chunks of 4
        calls_dec = list(diff.vae.calls)
        assert diff.encode(px) is px                           # pre-encoded: identity (:917-919)
        diff.pre_encoded = False
        diff.vae.calls.clear()
        dec_raw = diff.decode(z, chunk_size=20)
        enc = diff.encode(px, chunk_size=3)                    # 10 frames in chunks of 3
        calls_enc = [c for c in diff.vae.calls if c[0] == "encode"]
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(OUT, "decode.npz"), z=z.numpy(), px=px.numpy(), dec_pre=dec_pre.numpy(),
                        dec_raw=dec_raw.numpy(), enc=enc.numpy(), mean=st["mean"].numpy(), std=st["std"].numpy(),
                        dec_chunks=np.array([n for _, n in calls_dec]), enc_chunks=np.array([n for _, n in calls_enc]))
    print("[decode] ok", dec_pre.shape, enc.shape, calls_dec, calls_enc)


def gen_forward_cfgE_T20():
    """BASELINE.json configs[4] at FULL size (pixel space 128x128x3, 20 frames, batch 1, num_channels=128, the reference's
    default num_res_blocks=2 / channel_mult (1,1,2,3,4) / attention at 16x16 and 8x8).  The output (3.9 MB) is stored as a
    4x4-strided subsample plus per-(frame, channel) sums and L2 norms of the FULL output, so the fixture stays < 1 MB."""
    kw = CONFIGS["cfgE_T2"][0]
    cfg = uo.make_cfg(**kw)
    model, sd = build_reference_model(cfg)
    inp = recipe.make_inputs("cfgE_T20", 1, 20, cfg["in_channels"], 128, 128, n_pad=3)
    ti = tt(inp)
    ts = ti["t"].float()
    with torch.no_grad():
        ref, _ = model(ti["x"], x0=ti["x0"], timesteps=ts, frame_indices=ti["frame_indices"], obs_mask=ti["obs_mask"],
                       latent_mask=ti["latent_mask"])
        mine, _ = uo.unet_forward(sd, cfg, ti["x"], ti["x0"], ts, ti["frame_indices"], ti["obs_mask"], ti["latent_mask"])
    d = maxdiff(ref, mine)
    print(f"[forward cfgE_T20] out rms {float(ref.pow(2).mean().sqrt()):.4f}  oracle-vs-ref max|d| {d:.3e}")
    assert d < 2e-4
    r64 = ref.double()
    np.savez_compressed(os.path.join(OUT, "forward_cfgE_T20.npz"), sub=ref[..., ::4, ::4].numpy(),
                        frame_sum=r64.sum(dim=(-1, -2)).numpy(), frame_norm=r64.pow(2).sum(dim=(-1, -2)).sqrt().numpy(),
                        absmax=np.float64(ref.abs().max()), frame_indices=inp["frame_indices"],
                        n_params=np.int64(sum(v.numel() for v in sd.values())))


def gen_sampler_cfgB():
    """Three ancestral steps of the reference's p_sample from the top of the 1000-step chain at BASELINE.json configs[1]
    (ch64, batch 2, 20 frames, 4x16x16), with the recorded recipe noise: what the replayed cfg-B sampler plan - tune
    cache, timestep tables, GroupNorm epilogues - is compared with."""
    kw, B, T, H, n_pad = CONFIGS["cfgB"]
    cfg = uo.make_cfg(**kw)
    model, sd = build_reference_model(cfg)
    inp = tt(recipe.make_inputs("cfgB", B, T, cfg["in_channels"], H, H, n_pad=n_pad))
    shape = inp["x"].shape
    pixel = {"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None}
    diff = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing="", rescale_timesteps=True, rescale_learned_sigmas=True,
                                         diffusion_space_kwargs=dict(pixel))
    mk = dict(frame_indices=inp["frame_indices"], obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"], x0=inp["x0"])
    noise = [torch.from_numpy(recipe.gaussianish(f"samplerB/noise{i}", inp["x"].numel()).reshape(shape).astype(np.float32))
             for i in range(3)]
    real_randn_like = torch.randn_like
    x, traj = inp["x"].clone(), []
    for j, i in enumerate(range(999, 996, -1)):
        torch.randn_like = lambda x_, _n=noise[j]: _n
        try:
            with torch.no_grad():
                x = diff.p_sample(model, x, torch.tensor([i] * B), clip_denoised=True, model_kwargs=mk)["sample"]
        finally:
            torch.randn_like = real_randn_like
        traj.append(x.numpy())
    np.savez_compressed(os.path.join(OUT, "sampler_cfgB.npz"), traj=np.stack(traj))
    print("[sampler cfgB] ok", np.stack(traj).shape)


def gen_sampler_cfgD_window():
    """The long-video path (BASELINE.json configs[3]): batch 1, the configs[1] network, 250-step respacing, windows of the
    reference's hierarchy-2 schedule for T=1000 / 36 observed / K=20 / step 10 (tests/golden/schemes.json, itself generated
    from the reference's sampling_schemes): window 2 (20 frames: 10 conditioning frames incl. far-away anchors + 10 new) and
    the schedule's single 14-frame window.  Per window: three ancestral steps of the reference's SpacedDiffusion.p_sample
    from the top of the respaced chain (i = 249, 248, 247 -> model timesteps 999, 995, 991 through _WrappedModel) and the
    last two steps (i = 1, 0: the t == 0 step adds no noise), with recorded recipe noise (gaussian_diffusion.py:369-401,
    respace.py:110-124)."""
    import json
    kw, _, _, H, _ = CONFIGS["cfgB"]
    cfg = uo.make_cfg(**kw)
    model, sd = build_reference_model(cfg)
    with open(os.path.join(OUT, "schemes.json")) as f:
        case = next(c for c in json.load(f) if (c["scheme"], c["video_length"], c["n_obs"], c["max_frames"], c["step_size"])
                    == ("hierarchy-2", 1000, 36, 20, 10))
    wins = {20: 2, 14: next(i for i, w in enumerate(case["windows"]) if len(w[0]) + len(w[1]) == 14)}
    pixel = {"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None}
    diff = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing="250", rescale_timesteps=True, rescale_learned_sigmas=True,
                                         diffusion_space_kwargs=dict(pixel))
    assert diff.num_timesteps == 250
    out = {}
    real_randn_like = torch.randn_like
    for K, wi in wins.items():
        obs_idx, lat_idx = case["windows"][wi]
        assert len(obs_idx) + len(lat_idx) == K
        tag = f"cfgD_w{K}"
        inp = tt(recipe.make_inputs(tag, 1, K, cfg["in_channels"], H, H))
        fi = torch.tensor([list(obs_idx) + list(lat_idx)], dtype=torch.long)      # observed first (video_sample.py:56-61)
        obs = torch.zeros(1, K, 1, 1, 1)
        obs[:, :len(obs_idx)] = 1.0
        mk = dict(frame_indices=fi, obs_mask=obs, latent_mask=1 - obs, x0=inp["x0"])
        shape = inp["x"].shape
        for leg, steps, x in (("top", (249, 248, 247), inp["x"].clone()),
                              ("bottom", (1, 0), (0.5 * inp["x"] + 0.5 * inp["x0"]).clone())):
            traj = []
            for j, i in enumerate(steps):
                noise = torch.from_numpy(recipe.gaussianish(f"{tag}/{leg}/noise{j}", x.numel()).reshape(shape).astype(np.float32))
                torch.randn_like = lambda x_, _n=noise: _n
                try:
                    with torch.no_grad():
                        x = diff.p_sample(model, x, torch.tensor([i]), clip_denoised=True, model_kwargs=mk)["sample"]
                finally:
                    torch.randn_like = real_randn_like
                traj.append(x.numpy())
            out[f"w{K}_{leg}"] = np.stack(traj)
        out[f"w{K}_frame_indices"] = fi.numpy()
        out[f"w{K}_n_obs"] = np.int64(len(obs_idx))
        out[f"w{K}_window"] = np.int64(wi)
    np.savez_compressed(os.path.join(OUT, "sampler_cfgD_window.npz"), **out)
    print("[sampler cfgD window] ok", {k: v.shape for k, v in out.items()})


def gen_train_step_cfgC():
    """One optimizer step of the reference's TrainLoop arithmetic at BASELINE.json configs[2] (ch128, batch 2, 20 frames):
    training_losses (MSE branch, gaussian_diffusion.py:722-796) -> (losses["loss"] * weights).mean().backward()
    (train_util.py:320-328) -> AdamW(lr 1e-4, weight_decay 0) (train_util.py:103,346-351) -> update_ema at 0.9999
    (nn.py:55-65).  train_util.py itself cannot be imported here (mpi4py / blobfile are absent), so its three lines of
    arithmetic are driven directly with the reference's own model, diffusion, optimizer class and update_ema.
    30.5 M parameters: per-tensor summaries are stored (update norm, leading elements of gradient / new value / EMA)."""
    kw, B, T, H, n_pad = CONFIGS["cfgC"]
    cfg = uo.make_cfg(**kw)
    model, sd = build_reference_model(cfg)
    model.train()
    inp = tt(recipe.make_inputs("cfgC", B, T, cfg["in_channels"], H, H, n_pad=n_pad))
    pixel = {"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None}
    diff = rsu.create_gaussian_diffusion(steps=1000, timestep_respacing="", rescale_timesteps=True, rescale_learned_sigmas=True,
                                         diffusion_space_kwargs=dict(pixel))
    t = torch.tensor([700, 123])
    noise = torch.from_numpy(recipe.gaussianish("trainC/noise", inp["x0"].numel()).reshape(inp["x0"].shape).astype(np.float32))
    mk = dict(frame_indices=inp["frame_indices"], obs_mask=inp["obs_mask"], latent_mask=inp["latent_mask"], x0=inp["x0"])
    opt = torch.optim.AdamW(list(model.parameters()), lr=1e-4, weight_decay=0.0)
    ema = [p.detach().clone() for p in model.parameters()]
    old = [p.detach().clone() for p in model.parameters()]
    losses = diff.training_losses(model, inp["x0"], t, model_kwargs=mk, noise=noise, latent_mask=1 - inp["obs_mask"],
                                  eval_mask=inp["latent_mask"])
    weights = torch.ones(B)
    (losses["loss"] * weights).mean().backward()
    grads = [p.grad.detach().clone() for p in model.parameters()]
    opt.step()
    rnn.update_ema(ema, list(model.parameters()), rate=0.9999)
    keys = [n for n, _ in model.named_parameters()]
    new = [p.detach() for p in model.parameters()]
    head = lambda ts_: np.stack([np.resize(x.flatten()[:16].numpy(), 16) for x in ts_])
    np.savez_compressed(
        os.path.join(OUT, "train_step_cfgC.npz"), keys=np.array(keys), t=t.numpy(),
        loss=losses["loss"].detach().numpy(), mse=losses["mse"].detach().numpy(),
        grad_norm=np.array([float(g.double().norm()) for g in grads]), grad_head=head(grads),
        delta_norm=np.array([float((a - b).double().norm()) for a, b in zip(new, old)]),
        new_head=head(new), ema_head=head(ema), grad_absmax=np.array([float(g.abs().max()) for g in grads]))
    print("[train step cfgC] loss", losses["loss"].detach().numpy(), "total grad norm",
          float(np.sqrt(sum(float(g.double().pow(2).sum()) for g in grads))))


SCHEME_CASES = [
    # (scheme, video_length, n_obs, max_frames, step_size)
    ("autoreg", 1000, 36, 20, 10), ("autoreg", 1000, 0, 20, 10), ("autoreg", 47, 3, 8, 3),
    ("long-range", 1000, 36, 20, 10), ("long-range", 1000, 0, 20, 10), ("long-range", 61, 5, 10, 4),
    ("hierarchy-2", 1000, 36, 20, 10), ("hierarchy-2", 1000, 0, 20, 10), ("hierarchy-2", 300, 36, 14, 7),
    ("hierarchy-2", 100, 2, 8, 2), ("hierarchy-3", 1000, 36, 20, 10), ("hierarchy-3", 300, 0, 20, 5),
    ("hierarchy-4", 300, 5, 10, 4), ("hierarchy-5", 1000, 1, 20, 10),
]


def gen_schemes():
    """Window index sequences of the reference's non-adaptive sampling schemes."""
    import contextlib
    import io
    import json
    from improved_diffusion import sampling_schemes as rss  # reference
    cases = []
    for name, T, n_obs, K, step in SCHEME_CASES:
        with contextlib.redirect_stdout(io.StringIO()):
            it = iter(rss.sampling_schemes[name](video_length=T, num_obs=n_obs, max_frames=K, step_size=step))
            it.set_videos([None, None])          # batch of 2: every video gets the same lists
            windows = [([int(i) for i in o[0]], [int(i) for i in l[0]]) for o, l in it]
        assert sorted(set(range(n_obs)) | {i for _, l in windows for i in l}) == list(range(T))
        cases.append(dict(scheme=name, video_length=T, n_obs=n_obs, max_frames=K, step_size=step, windows=windows))
        print(f"[schemes] {name} T={T} n_obs={n_obs} K={K} step={step}: {len(windows)} windows")
    with open(os.path.join(OUT, "schemes.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "schemes":
        gen_schemes()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "forward":
        gen_forward(only=sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "backward":
        gen_backward(names=sys.argv[2:])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "decode":
        gen_decode()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ops":
        gen_ops()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round4":       # the fixtures added in round 4 only
        gen_forward(only=["micro_px"])
        gen_backward(names=["micro_px", "cfgE_T2"])
        gen_sampler_cfgD_window()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfgD_window":
        gen_sampler_cfgD_window()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round3":       # the fixtures added in round 3 only
        gen_sampler_cfgB()
        gen_train_step_cfgC()
        gen_forward_cfgE_T20()
        sys.exit(0)
    gen_ops()
    gen_forward()
    gen_backward()
    gen_diffusion()
    gen_decode()
    gen_schemes()
    gen_sampler_cfgB()
    gen_train_step_cfgC()
    gen_forward_cfgE_T20()
    gen_sampler_cfgD_window()
    print("golden vectors written to", OUT)
