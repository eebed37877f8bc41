"""CPU-only checks of the host side: module tree / state-dict ABI, factories, diffusion tables against the
reference's golden vectors, respacing, the C-ABI library exports, mask sampling invariants, samplers."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import unet_oracle as uo
from conftest import GOLDEN, ROOT
from test_oracle_golden import CONFIGS


def native_model(cfg):
    from improved_diffusion.unet import UNetVideoModel
    return UNetVideoModel(in_channels=cfg["in_channels"], model_channels=cfg["model_channels"], out_channels=cfg["out_channels"],
                          num_res_blocks=cfg["num_res_blocks"], attention_resolutions=cfg["attention_resolutions"],
                          channel_mult=cfg["channel_mult"], num_heads=cfg["num_heads"], use_scale_shift_norm=True,
                          use_rpe_net=True)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_state_dict_abi_matches_reference(name):
    """Key names, shapes and named_parameters() ORDER equal the reference's (pinned in make_golden.py)."""
    cfg = uo.make_cfg(**CONFIGS[name][0])
    m = native_model(cfg)
    want = uo.param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got.keys()) == list(want.keys())
    assert got == {k: tuple(v) for k, v in want.items()}
    assert [k for k, _ in m.named_parameters()] == list(want.keys())
    assert len(list(m.buffers())) == 0
    g = np.load(os.path.join(GOLDEN, f"forward_{name}.npz"))
    assert int(g["n_params"]) == sum(p.numel() for p in m.parameters())
    # zero-initialised layers of the reference (unet.py:168-170,402; rpe.py:14-16,112)
    sd = m.state_dict()
    for k, v in sd.items():
        if re.search(r"(out_layers\.3|proj_out|rpe_net\.out|^out\.2)\.(weight|bias)$", k):
            assert float(v.abs().max()) == 0.0, k


def test_cpu_forward_fails_loudly():
    cfg = uo.make_cfg(**CONFIGS["micro"][0])
    m = native_model(cfg)
    x = torch.zeros(1, 2, 4, 16, 16)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x, x0=x, timesteps=torch.zeros(1), frame_indices=torch.zeros(1, 2, dtype=torch.long),
          obs_mask=torch.zeros(1, 2, 1, 1, 1), latent_mask=torch.ones(1, 2, 1, 1, 1))


def test_factories_and_defaults():
    from improved_diffusion import script_util as su, gaussian_diffusion as gd
    d = su.model_and_diffusion_defaults()
    assert len(d) == 22 and d["num_channels"] == 128 and d["use_rpe_net"] and d["rescale_timesteps"]
    d.update(image_size=32, in_channels=4, num_channels=32, num_res_blocks=1, diffusion_steps=32,
             diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None})
    model, diff = su.create_model_and_diffusion(**d)
    assert model.channel_mult == (1, 2, 2, 2) and model.attention_resolutions == (2, 4)   # image_size // (16, 8)
    assert sum(p.numel() for p in model.parameters()) == 2063268                            # SURVEY §8c [measured]
    assert diff.num_timesteps == 32 and diff.loss_type == gd.LossType.RESCALED_MSE
    assert diff.model_mean_type == gd.ModelMeanType.EPSILON and diff.model_var_type == gd.ModelVarType.FIXED_LARGE
    with pytest.raises(ValueError):
        su.create_model(48, 4, 32, 1, False, False, False, "16,8", 4, -1, True, 0.0, True)
    # declared extension: 16x16 latents
    m16 = su.create_model(16, 4, 64, 1, False, False, False, "16,8", 4, -1, True, 0.0, True)
    assert m16.attention_resolutions == (1, 2) and sum(p.numel() for p in m16.parameters()) == 7645380
    p = argparse.ArgumentParser()
    su.add_dict_to_argparser(p, dict(a=1, flag=True, name=None))
    ns = p.parse_args(["--flag", "no", "--a", "3"])
    assert ns.a == 3 and ns.flag is False and su.args_to_dict(ns, ["a"]) == {"a": 3}
    assert su.str2bool("Yes") and not su.str2bool("0")


def test_diffusion_tables_bit_exact_vs_reference():
    from improved_diffusion import script_util as su, respace
    g = np.load(os.path.join(GOLDEN, "diffusion.npz"))
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    pixel = {"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None}
    for tag, steps, resp in (("lin1000", 1000, ""), ("lin32", 32, ""), ("lin1000_r250", 1000, "250"), ("cos100_r10_15", 100, "10,15")):
        diff = su.create_gaussian_diffusion(steps=steps, noise_schedule="cosine" if tag.startswith("cos") else "linear",
                                            timestep_respacing=resp, rescale_timesteps=True, rescale_learned_sigmas=True,
                                            diffusion_space_kwargs=dict(pixel))
        for n in names:
            assert np.array_equal(getattr(diff, n), g[f"{tag}/{n}"]), (tag, n)
        assert np.array_equal(np.array(diff.timestep_map), g[f"{tag}/timestep_map"])
    assert respace.space_timesteps(300, "10,15,20") == respace.space_timesteps(300, [10, 15, 20])
    assert len(respace.space_timesteps(1000, "ddim50")) == 50
    with pytest.raises(ValueError):
        respace.space_timesteps(10, "20")
    w = respace._WrappedModel(lambda x, timesteps, **k: timesteps, [0, 4, 8], True, 1000)
    assert torch.equal(w(None, torch.tensor([2, 0])), torch.tensor([8.0, 0.0]))


def test_c_abi_exports_every_declared_symbol():
    """The library loads without a GPU and exports every entry point include/lfvdm_hip.h declares."""
    from improved_diffusion import _native
    hdr = open(os.path.join(ROOT, "include", "lfvdm_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lfvdm_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert os.path.exists(_native.LIB_PATH), "run `python __graft_entry__.py` (build) first"
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    for name in _native.EXPORTS:
        assert name in declared, f"{name} is bound in Python but not declared in the header"
    assert lib.lfvdm_abi_version() == 9


def _loop_stub(max_frames=6, pad=True):
    from improved_diffusion.train_util import TrainLoop
    loop = TrainLoop.__new__(TrainLoop)
    loop.max_frames, loop.pad_with_random_frames = max_frames, pad
    return loop


def test_mask_sampling_invariants():
    """sample_all_masks / prepare_training_batch (reference train_util.py:193-241): at most max_frames frames
    flagged, obs and latent disjoint, gathered indices sorted then random padding."""
    torch.manual_seed(0); np.random.seed(0)
    loop = _loop_stub()
    B, T = 5, 30
    batch1 = torch.arange(B * T, dtype=torch.float32).view(B, T, 1, 1, 1).expand(B, T, 2, 3, 3).contiguous()
    batch2 = -batch1
    for _ in range(20):
        _, obs_full, lat_full = loop.sample_all_masks(batch1, batch2, gather=False)
        assert float((obs_full * lat_full).sum()) == 0.0                      # disjoint
        any_mask = (obs_full + lat_full).clip(max=1)
        k_all = any_mask.view(B, T).sum(1).long()
        assert (k_all >= 1).all() and (k_all <= 6).all()                      # <= max_frames flagged
        batch, (obs, lat), fi = loop.prepare_training_batch(any_mask, batch1, batch2, (obs_full, lat_full))
        assert batch.shape == (B, 6, 2, 3, 3) and fi.shape == (B, 6) and obs.shape == (B, 6, 1, 1, 1)
        for b in range(B):
            k = int(k_all[b])
            assert torch.equal(fi[b, :k], any_mask[b].view(T).nonzero().flatten())   # flagged frames, sorted
            assert torch.equal(batch[b, :k, 0, 0, 0], batch1[b, fi[b, :k], 0, 0, 0])
            assert torch.equal(batch[b, k:, 0, 0, 0], batch2[b, fi[b, k:], 0, 0, 0])   # random padding from batch2
            # masks are gathered at the same indices (padding frames inherit the mask of their source index:
            # reference train_util.py:238-240)
            assert torch.equal(obs[b].view(-1), obs_full[b].view(T)[fi[b]]) and torch.equal(lat[b].view(-1), lat_full[b].view(T)[fi[b]])
    idx = loop.sample_some_indices(6, 30)
    assert 1 <= len(idx) <= 6 and all(0 <= i < 30 for i in idx)
    loop2 = _loop_stub(pad=False)
    batch, fi, obs, lat = loop2.sample_all_masks(batch1, None)
    assert batch.shape[1] == int((obs + lat).view(B, -1).sum(1).max())


def test_sample_some_indices_equals_the_reference_formula():
    """The vectorised float32 evaluation in TrainLoop.sample_some_indices gives exactly the indices of the reference's
    element-by-element ``int(pos + i*scale)`` with a float32 0-dim tensor ``pos`` (train_util.py:180-191)."""
    def reference_formula(max_indices, T):
        while True:
            s = int(torch.randint(low=1, high=max_indices + 1, size=()))
            max_scale = T / (float(s) - 0.999)
            scale = np.exp(np.random.rand() * np.log(max_scale))
            pos = torch.rand(()) * (T - scale * (s - 1))
            indices = [int(pos + i * scale) for i in range(s)]
            if all(0 <= i < T for i in indices):
                return indices
    loop = _loop_stub()
    for N, T in ((20, 40), (20, 1000), (6, 30), (3, 7)):
        torch.manual_seed(N + T); np.random.seed(N + T)
        want = [reference_formula(N, T) for _ in range(1500)]
        torch.manual_seed(N + T); np.random.seed(N + T)
        got = [loop.sample_some_indices(N, T) for _ in range(1500)]
        assert got == want


def test_index_table_sampler_invariants_and_distribution():
    """The index table of the device-side batch preparation (TrainLoop.sample_index_table): the same
    invariants as the host sampler above, bit-identical batches from the same seed, and - from independent seeds - the
    same distribution of the number of flagged frames over 10 000 draws (two-sample chi-square against the host sampler;
    the two samplers run the same random process, so this guards refactors of either one)."""
    loop = _loop_stub(max_frames=6)
    B, T = 4, 30
    batch1 = torch.arange(B * T, dtype=torch.float32).view(B, T, 1, 1, 1).expand(B, T, 1, 2, 2).contiguous()
    batch2 = -batch1 - 1
    pool = torch.cat([batch1, batch2], dim=1)
    for seed in range(10):
        torch.manual_seed(seed); np.random.seed(seed)
        want = loop.sample_all_masks(batch1, batch2)
        torch.manual_seed(seed); np.random.seed(seed)
        tab = loop.sample_index_table(B, T)
        assert tab.shape == (B, 6, 4) and tab.dtype == np.int32
        got = torch.stack([pool[b][torch.from_numpy(tab[b, :, 0]).long()] for b in range(B)])
        assert torch.equal(got, want[0]) and np.array_equal(tab[:, :, 1], want[1].numpy())
        assert np.array_equal(tab[:, :, 2], want[2].view(B, -1).numpy()) and np.array_equal(tab[:, :, 3], want[3].view(B, -1).numpy())
        for row in tab:
            n = int((row[:, 0] < T).sum())
            assert 1 <= n <= 6 and (row[:n, 0] < T).all() and (row[n:, 0] >= T).all()            # flagged first, then padding
            assert (np.diff(row[:n, 1]) > 0).all() and ((row[:n, 2] + row[:n, 3]) == 1).all()    # sorted; obs xor latent
            assert ((row[:, 2] + row[:, 3]) <= 1).all()
    draws = 2500                      # x B = 10 000 videos per sampler
    torch.manual_seed(1000); np.random.seed(1000)
    h_table = np.zeros(7)
    for _ in range(draws):
        tab = loop.sample_index_table(B, T)
        for row in tab:
            h_table[int((row[:, 0] < T).sum())] += 1
    torch.manual_seed(2000); np.random.seed(2000)
    h_host = np.zeros(7)
    for _ in range(draws):
        _, obs_full, lat_full = loop.sample_all_masks(batch1, batch2, gather=False)
        for k in (obs_full + lat_full).view(B, T).sum(1).long().tolist():
            h_host[k] += 1
    assert h_table.sum() == h_host.sum() == draws * B and h_table[0] == h_host[0] == 0
    keep = (h_table + h_host) > 0
    chi2 = float((((h_table - h_host) ** 2) / (h_table + h_host))[keep].sum())
    assert chi2 < 27.9, (chi2, h_table, h_host)          # chi-square, 6 degrees of freedom, p = 1e-4


def test_samplers_logger_rng_helpers():
    from improved_diffusion import resample, logger as lg, rng_util, train_util
    class D: num_timesteps = 50
    s = resample.create_named_schedule_sampler("uniform", D())
    np.random.seed(0)
    t, w = s.sample(8, torch.device("cpu"))
    assert t.dtype == torch.long and t.shape == (8,) and torch.allclose(w, torch.ones(8))
    r = resample.create_named_schedule_sampler("loss-second-moment", D())
    r.update_with_all_losses([1, 1, 2], [0.5, 0.25, 1.0])
    assert r.weights().shape == (50,) and not r._warmed_up()
    with pytest.raises(NotImplementedError):
        resample.create_named_schedule_sampler("nope", D())
    L = lg.Logger()
    L.logkv("a", 1.0); L.logkv_mean("b", 2.0); L.logkv_mean("b", 4.0)
    out = L.dumpkvs()
    assert out["a"] == 1.0 and out["b"] == 3.0 and len(L.name2val) == 0
    with rng_util.RNG(3):
        a = torch.rand(2)
    with rng_util.RNG(3):
        b = torch.rand(2)
    assert torch.equal(a, b)
    # a private stream continues where it stopped and never disturbs the caller's generators
    torch.manual_seed(5); np.random.seed(5)
    want_outer = (torch.rand(3), np.random.rand(3))
    torch.manual_seed(5); np.random.seed(5)
    priv = rng_util.RNG(9)
    with priv:
        p1 = (torch.rand(2), np.random.rand(2))
    with priv:
        p2 = (torch.rand(2), np.random.rand(2))
    got_outer = (torch.rand(3), np.random.rand(3))
    assert torch.equal(got_outer[0], want_outer[0]) and np.array_equal(got_outer[1], want_outer[1])
    with rng_util.RNG(9):
        q = (torch.rand(4), np.random.rand(4))
    assert torch.equal(torch.cat([p1[0], p2[0]]), q[0]) and np.array_equal(np.concatenate([p1[1], p2[1]]), q[1])
    @rng_util.rng_decorator(seed=4)
    def draw():
        return torch.rand(2)
    assert torch.equal(draw(), draw())
    # loss-aware sampler: uniform until every timestep has a full history, then RMS of the kept losses (+ uniform mix)
    class D3: num_timesteps = 3
    r3 = resample.LossSecondMomentResampler(D3(), history_per_term=2, uniform_prob=0.1)
    r3.update_with_local_losses(torch.tensor([0, 1, 2, 0]), torch.tensor([1.0, 2.0, 2.0, 3.0]))
    assert np.array_equal(r3.weights(), np.ones(3))
    r3.update_with_all_losses([1, 2, 0], [2.0, 4.0, 5.0])          # t=0 now keeps (3, 5): the 1.0 fell out
    rms = np.sqrt(np.array([(9 + 25) / 2, 4.0, (4 + 16) / 2]))
    assert np.allclose(r3.weights(), rms / rms.sum() * 0.9 + 0.1 / 3)
    assert train_util.parse_resume_step_from_filename("x/model012345.pt") == 12345
    assert train_util.parse_resume_step_from_filename("x/ema.pt") == 0
    class Diff: num_timesteps = 100
    lg.logger.name2val.clear(); lg.logger.name2cnt.clear()
    train_util.log_loss_dict(Diff(), torch.tensor([10, 90]), {"loss": torch.tensor([1.0, 3.0])})
    assert lg.logger.name2val["loss"] == 2.0 and lg.logger.name2val["loss_q0"] == 1.0 and lg.logger.name2val["loss_q3"] == 3.0
    lg.logger.dumpkvs()


def test_update_ema_and_zero_grad_helpers():
    from improved_diffusion.nn import update_ema, mean_flat, timestep_embedding
    from improved_diffusion.fp16_util import zero_grad
    src = [torch.ones(3), torch.full((2, 2), 2.0)]
    tgt = [torch.zeros(3), torch.zeros(2, 2)]
    update_ema(tgt, src, rate=0.9)
    assert torch.allclose(tgt[0], torch.full((3,), 0.1)) and torch.allclose(tgt[1], torch.full((2, 2), 0.2))
    p = torch.nn.Parameter(torch.ones(2)); p.grad = torch.ones(2)
    zero_grad([p])
    assert float(p.grad.abs().sum()) == 0.0
    x = torch.arange(8.0).view(2, 2, 2)
    assert torch.allclose(mean_flat(x, torch.tensor([[[1.0], [0.0]]])), torch.tensor([0.25, 2.25]))
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    assert np.array_equal(timestep_embedding(torch.from_numpy(g["temb_t"]), 64).numpy(), g["temb_64"])


def test_vae_boundary_matches_reference_golden():
    """encode / decode around the (here: stand-in) frame autoencoder against outputs of the REFERENCE's encode / decode
    run with the same stand-in (oracle/make_golden.py gen_decode; reference gaussian_diffusion.py:914-947): pre-encoded
    de-normalisation with a stats dict, chunking, reshapes.  Not a HIP path: plain tensor plumbing, so it runs here."""
    from improved_diffusion import script_util as su
    from oracle import fake_vae
    g = np.load(os.path.join(GOLDEN, "decode.npz"))
    z, px = torch.from_numpy(g["z"]), torch.from_numpy(g["px"])
    st = {"mean": torch.from_numpy(g["mean"]), "std": torch.from_numpy(g["std"])}      # as torch.load(stats file) gives it
    diff = su.create_gaussian_diffusion(steps=1000, diffusion_space_kwargs={
        "diffusion_space": "latent", "pre_encoded": True, "pre_encoded_stats_dict": st})
    assert diff.vae is None and diff.pre_encoded_stats_dict["std"].shape == (1, 1, 4, 1, 1)
    # no autoencoder attached: pre-encoded latents come back de-normalised (what the decoder would be fed)
    want = z * st["std"].view(1, 1, 4, 1, 1) + st["mean"].view(1, 1, 4, 1, 1)
    with pytest.raises(NotImplementedError):         # the reference always returns pixels: no silent latents
        diff.decode(z)
    assert not diff.can_decode()
    assert torch.equal(diff.decode(z, allow_latents=True), want) and torch.equal(diff.denormalize_latents(z), want)
    assert diff.encode(px) is px
    # with the stand-in attached: the reference's outputs, chunk for chunk
    vae = fake_vae.FakeVAE()
    diff.set_vae(vae, fake_vae.FakeImageProcessor(), dtype=torch.float32)
    np.testing.assert_array_equal(diff.decode(z, chunk_size=4).numpy(), g["dec_pre"])
    assert [n for k, n in vae.calls if k == "decode"] == list(g["dec_chunks"])
    raw = su.create_gaussian_diffusion(steps=1000, diffusion_space_kwargs={
        "diffusion_space": "latent", "pre_encoded": False, "pre_encoded_stats_dict": None})
    with pytest.raises(NotImplementedError):
        raw.decode(z)
    with pytest.raises(NotImplementedError):
        raw.decode(z, allow_latents=True)            # not pre-encoded: there is nothing meaningful to hand back
    with pytest.raises(NotImplementedError):
        raw.encode(px)
    vae2 = fake_vae.FakeVAE()
    raw.set_vae(vae2, fake_vae.FakeImageProcessor(), dtype=torch.float32)
    enc = raw.encode(px, chunk_size=3)
    np.testing.assert_array_equal(enc.numpy(), g["enc"])
    assert [n for k, n in vae2.calls if k == "encode"] == list(g["enc_chunks"])
    np.testing.assert_array_equal(raw.decode(z, chunk_size=20).numpy(), g["dec_raw"])
    # pixel space: both are the identity
    pix = su.create_gaussian_diffusion(steps=1000, diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False,
                                                                         "pre_encoded_stats_dict": None})
    assert pix.decode(z) is z and pix.encode(px) is px


def test_wgrad_tune_codes_are_well_formed():
    """Candidate launch codes of the weight-gradient tuner (_native._wgrad_codes: 1 + tile + 4 * stages + 16 * M slices; tile
    1 / 2 / 3 = 64x64 / 128x64 / 128x128, stages 1 / 2 / 3 = two / three stages / 64-row chunks, stage field 0 = the tap-fused
    kernels with three (tile 1) or nine (tile 2) taps per workgroup): unique, only legal tiles for the filter / channel
    count, a bounded number of M slices, tap-fused codes only for 3x3 / stride-1 layers on maps of 8, 16 or a multiple of
    32 pixels per row, none at all for shapes the LDS-DMA kernels do not take."""
    from improved_diffusion import _native as nat
    a = nat.ConvArgs()
    a.N, a.Hs, a.Ws, a.Ho, a.Wo, a.C0, a.C1, a.Cout, a.ksize, a.stride = 40, 16, 16, 16, 16, 128, 0, 128, 3, 1
    codes = nat._wgrad_codes(a)
    assert codes and len(set(codes)) == len(codes)
    nchunks = (40 * 256 + 31) // 32
    seen = set()
    for c in codes:
        t = c - 1
        tile, stages, ms = t & 3, (t >> 2) & 3, t >> 4
        assert tile in (1, 2, 3) and 1 <= ms <= nchunks // 2
        assert stages in (1, 2, 3) or (stages == 0 and tile in (1, 2)), (tile, stages)
        seen.add((tile, stages > 0))
    assert seen == {(1, True), (2, True), (3, True), (1, False), (2, False)}
    a.stride = 2                                              # strided layer: no tap-fused kernels
    assert all(((c - 1) >> 2) & 3 for c in nat._wgrad_codes(a))
    a.stride, a.Ho, a.Wo, a.Hs, a.Ws = 1, 12, 12, 12, 12       # 144 pixels per frame: chunks would straddle frames
    assert all(((c - 1) >> 2) & 3 for c in nat._wgrad_codes(a))
    a.Ho = a.Wo = a.Hs = a.Ws = 16
    a.Cout = 64
    assert {(c - 1) & 3 for c in nat._wgrad_codes(a) if ((c - 1) >> 2) & 3} == {1}, "128-filter tiles need 128 filters"
    a.C0 = 96
    assert nat._wgrad_codes(a) == []
    a.C0, a.Cout = 128, 32
    assert nat._wgrad_codes(a) == []

