"""Pins the CPU oracle (oracle/*.py) to golden vectors produced by the REAL reference
(oracle/make_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe, unet_oracle as uo, diffusion_oracle as do

from conftest import GOLDEN

CONFIGS = {
    "micro": (dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)), 2, 4, 16, 1),
    "micro_rb2": (dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(2,), num_res_blocks=2,
                       num_heads=2), 1, 3, 8, 0),
    "cfgA": (dict(model_channels=32, channel_mult=(1, 2, 2, 2), attention_resolutions=(2, 4)), 1, 5, 32, 0),
    # pixel-like micro model: 3 input / 3 output channels (the head's Cout is not a multiple of 4)
    "micro_px": (dict(in_channels=3, model_channels=64, channel_mult=(1, 2), attention_resolutions=(2,)), 1, 3, 16, 0),
    "cfgB": (dict(model_channels=64, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 20, 16, 3),
    "cfgB_T14": (dict(model_channels=64, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 14, 16, 0),
    # BASELINE.json configs[2]: per-GPU training workload (ch128, 4 levels, 20 frames incl. 3 padding frames, batch 2)
    "cfgC": (dict(model_channels=128, channel_mult=(1, 2, 2, 2), attention_resolutions=(1, 2)), 2, 20, 16, 3),
    # BASELINE.json configs[4]: pixel space 128x128x3, num_channels=128, num_res_blocks=2 (head dims 96 / 128), 2 frames
    "cfgE_T2": (dict(in_channels=3, model_channels=128, num_res_blocks=2, channel_mult=(1, 1, 2, 3, 4),
                     attention_resolutions=(8, 16)), 1, 2, 128, 0),
}


def tt(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def load_case(name):
    kw, B, T, H, n_pad = CONFIGS[name]
    cfg = uo.make_cfg(**kw)
    sd = tt(recipe.fill_state_dict(uo.param_shapes(cfg)))
    inp = tt(recipe.make_inputs(name, B, T, cfg["in_channels"], H, H, n_pad=n_pad))
    return cfg, sd, inp


@pytest.mark.parametrize("name", list(CONFIGS))
def test_forward_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"forward_{name}.npz"))
    cfg, sd, inp = load_case(name)
    assert int(g["n_params"]) == sum(v.numel() for v in sd.values())
    assert abs(float(g["x_sum"]) - float(inp["x"].double().sum())) < 1e-6
    with torch.no_grad():
        out, attn = uo.unet_forward(sd, cfg, inp["x"], inp["x0"], inp["t"].float(), inp["frame_indices"],
                                    inp["obs_mask"], inp["latent_mask"], return_attn_weights=True)
    # tolerance: fp32 re-association noise; the reference itself is 4e-5..1.4e-4 away from fp64
    np.testing.assert_allclose(out.numpy(), g["out"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(attn["temporal"][0].numpy()[:8], g["attn_t0"], atol=5e-5)
    np.testing.assert_allclose(attn["spatial"][0].numpy()[:1, :32, :32], g["attn_s0"], atol=5e-5)


@pytest.mark.parametrize("name", ["micro", "micro_rb2", "micro_px", "cfgC", "cfgE_T2"])
def test_backward_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"backward_{name}.npz"))
    cfg, sd, inp = load_case(name)
    probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = inp["x"].clone().requires_grad_(True)
    out, _ = uo.unet_forward(sd, cfg, x, inp["x0"], inp["t"].float(), inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])
    (out * probe).sum().backward()
    keys = [str(k) for k in g["keys"]]
    assert keys == list(sd.keys())
    gmax = float(g["gmax"])
    for i, k in enumerate(keys):
        gr = sd[k].grad
        tol = 2e-3 * (float(g["norms"][i]) + 1e-3 * gmax * np.sqrt(gr.numel()))
        assert abs(float(gr.double().norm()) - float(g["norms"][i])) < tol, k
        np.testing.assert_allclose(np.resize(gr.flatten()[:8].numpy(), 8), g["head"][i],
                                   atol=2e-3 * (np.abs(g["head"][i]).max() + 1e-3 * gmax), err_msg=k)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=2e-4 * np.abs(g["dx"]).max())


def test_ops_fixtures():
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    t = torch.from_numpy(g["temb_t"])
    assert np.array_equal(uo.timestep_embedding(t, 64).numpy(), g["temb_64"])
    # RPE bias einsum against the reference's scalar definition (rpe.py:85-96)
    B, D, Hh, T, Fh = 2, 3, 2, 5, 16
    qk = torch.from_numpy(recipe.gaussianish("ops/rpe/qk", B * D * Hh * T * Fh).reshape(B, D, Hh, T, Fh).astype(np.float32))
    temb = torch.from_numpy(recipe.gaussianish("ops/rpe/temb", B * T * 128).reshape(B * T, 128).astype(np.float32))
    fi = torch.from_numpy(g["rpe_fi"])
    rel = fi.unsqueeze(-1) - fi.unsqueeze(-2)
    shapes = {"embed_distances.weight": (32, 3), "embed_distances.bias": (32,),
              "embed_diffusion_time.weight": (32, 128), "embed_diffusion_time.bias": (32,),
              "out.weight": (32, 32), "out.bias": (32,)}
    sd = {"p." + k: torch.from_numpy(recipe.fill_param("ops/rpe.rpe_net." + k, s)) for k, s in shapes.items()}
    R = uo.rpe_net(sd, "p", temb, rel, 2)
    np.testing.assert_allclose(R.numpy(), g["rpe_R"], atol=1e-6)
    safe = torch.einsum("bdhtf,btshf->bdhts", qk, R)
    np.testing.assert_allclose(safe.numpy(), g["rpe_safe_qk"], atol=1e-5)
    # ResBlock with Cin != Cout
    rshapes = {"in_layers.0.weight": (64,), "in_layers.0.bias": (64,), "in_layers.2.weight": (32, 64, 3, 3),
               "in_layers.2.bias": (32,), "emb_layers.1.weight": (64, 128), "emb_layers.1.bias": (64,),
               "out_layers.0.weight": (32,), "out_layers.0.bias": (32,), "out_layers.3.weight": (32, 32, 3, 3),
               "out_layers.3.bias": (32,), "skip_connection.weight": (32, 64, 1, 1), "skip_connection.bias": (32,)}
    rsd = {"p." + k: torch.from_numpy(recipe.fill_param("ops/res." + k, s)) for k, s in rshapes.items()}
    x = torch.from_numpy(recipe.gaussianish("ops/res/x", 3 * 64 * 8 * 8).reshape(3, 64, 8, 8).astype(np.float32))
    emb = torch.from_numpy(recipe.gaussianish("ops/res/emb", 3 * 128).reshape(3, 128).astype(np.float32))
    np.testing.assert_allclose(uo.res_block(rsd, "p", x, emb).numpy(), g["res_y"], atol=2e-5)


def test_diffusion_tables_and_steps():
    g = np.load(os.path.join(GOLDEN, "diffusion.npz"))
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    for tag, steps, resp in (("lin1000", 1000, ""), ("lin32", 32, ""), ("lin1000_r250", 1000, "250"),
                             ("cos100_r10_15", 100, "10,15")):
        base = do.cosine_betas(steps) if tag.startswith("cos") else do.linear_betas(steps)
        tab = do.Tables(base, do.space_timesteps(steps, resp) if resp else None)
        for n in names:
            assert np.array_equal(getattr(tab, n), g[f"{tag}/{n}"]), (tag, n)  # bit-exact float64
        assert np.array_equal(np.array(tab.timestep_map), g[f"{tag}/timestep_map"])
    tm = g["lin1000_r250/timestep_map"]
    assert tm[0] == 0 and tm[1] == 4 and tm[-1] == 999 and len(tm) == 250
    # model-driven: q_sample / losses / trajectory on the micro model
    cfg, sd, inp = load_case("micro")
    shape = inp["x"].shape
    noise = [torch.from_numpy(recipe.gaussianish(f"diff/noise{i}", inp["x"].numel()).reshape(shape).astype(np.float32))
             for i in range(6)]

    def model_fn(x_t, ts):
        return uo.unet_forward(sd, cfg, x_t, inp["x0"], ts, inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])[0]

    for tag, steps, resp in (("lin1000", 1000, ""), ("lin1000_r250", 1000, "250")):
        tab = do.Tables(do.linear_betas(steps), do.space_timesteps(steps, resp) if resp else None)
        t = torch.from_numpy(g[f"{tag}/t"])
        with torch.no_grad():
            np.testing.assert_allclose(do.q_sample(tab, inp["x0"], t, noise[0]).numpy(), g[f"{tag}/q_sample"], atol=1e-6)
            losses = do.training_losses(tab, model_fn, inp["x0"], t, noise[0], 1 - inp["obs_mask"], inp["latent_mask"])
            for k, v in losses.items():
                np.testing.assert_allclose(v.numpy(), g[f"{tag}/loss/{k}"], rtol=1e-4, atol=1e-6)
            nt = tab.num_timesteps
            x = inp["x"].clone()
            for j, i in enumerate(range(nt - 1, nt - 6, -1)):
                ti = torch.tensor([i] * shape[0])
                x, _ = do.p_sample(tab, model_fn(x, do.model_timesteps(tab, ti)), x, ti, noise[j + 1])
                np.testing.assert_allclose(x.numpy(), g[f"{tag}/traj"][j], atol=1e-4 * (j + 1))
            ti = torch.zeros(shape[0], dtype=torch.long)
            x_last, _ = do.p_sample(tab, model_fn(inp["x"], do.model_timesteps(tab, ti)), inp["x"], ti, noise[0])
            np.testing.assert_allclose(x_last.numpy(), g[f"{tag}/p_sample_t0"], atol=1e-4)


@pytest.mark.parametrize("K", [20, 14])
def test_long_video_window_trajectory(K):
    """The oracle's respaced p_sample at the long-video shapes (batch 1, hierarchy-2 window frame indices, 250 steps)
    against the reference's trajectory (tests/golden/sampler_cfgD_window.npz)."""
    g = np.load(os.path.join(GOLDEN, "sampler_cfgD_window.npz"))
    cfg, sd, _ = load_case("cfgB")
    tag = f"cfgD_w{K}"
    inp = tt(recipe.make_inputs(tag, 1, K, cfg["in_channels"], 16, 16))
    fi = torch.from_numpy(g[f"w{K}_frame_indices"])
    obs = torch.zeros(1, K, 1, 1, 1)
    obs[:, :int(g[f"w{K}_n_obs"])] = 1.0
    tab = do.Tables(do.linear_betas(1000), do.space_timesteps(1000, "250"))
    shape = inp["x"].shape
    with torch.no_grad():
        for leg, steps, x in (("top", (249, 248, 247), inp["x"].clone()), ("bottom", (1, 0), 0.5 * inp["x"] + 0.5 * inp["x0"])):
            for j, i in enumerate(steps):
                noise = torch.from_numpy(recipe.gaussianish(f"{tag}/{leg}/noise{j}", x.numel()).reshape(shape).astype(np.float32))
                ti = torch.tensor([i])
                eps = uo.unet_forward(sd, cfg, x, inp["x0"], do.model_timesteps(tab, ti), fi, obs, 1 - obs)[0]
                x, _ = do.p_sample(tab, eps, x, ti, noise)
                np.testing.assert_allclose(x.numpy(), g[f"w{K}_{leg}"][j], atol=1e-4 * (j + 1))
