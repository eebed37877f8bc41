"""Windowed long-video sampling on the MI355X (BASELINE.json configs[3] in miniature): hierarchy-2 / autoreg
windows of varying length through the captured sampler, observed frames untouched, bitwise reproducible."""
import pytest
import torch

from test_oracle_golden import load_case
from test_forward_gpu import build_native
from test_sampler_gpu import make_diffusion
from test_long_video_cpu import run_scheme

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme,T,n_obs,K,step", [("hierarchy-2", 42, 4, 8, 4), ("autoreg", 21, 0, 6, 2)])
def test_sample_video_on_device(scheme, T, n_obs, K, step):
    from improved_diffusion.video_sampler import sample_video, default_sampling_args
    cfg, sd, inp = load_case("micro")
    model = build_native(cfg, sd).eval()
    diff = make_diffusion(1000, "3")          # 3 respaced steps per window keep the test short
    B = 2
    g = torch.Generator().manual_seed(5)
    batch = torch.randn(B, T, 4, 16, 16, generator=g)
    args = default_sampling_args(sampling_scheme=scheme, n_obs=n_obs, max_frames=K, max_latent_frames=step, device="cuda")
    outs = []
    for _ in range(2):
        torch.manual_seed(11)
        samples, used = sample_video(args, model, diff, batch, verbose=False)
        outs.append(samples)
    want = run_scheme(scheme, T, n_obs, K, step)
    assert [[list(map(int, o[0])), list(map(int, l[0]))] for o, l in used] == want
    lengths = {len(o) + len(l) for o, l in want}
    assert len(lengths) >= 2, "the schedule should exercise more than one window length"
    s = outs[0]
    assert s.shape == batch.shape and s.device == batch.device and torch.isfinite(s).all()
    assert torch.equal(s[:, :n_obs], batch[:, :n_obs])
    assert float(s[:, n_obs:].abs().max()) <= 1.0 + 1e-6            # clip_denoised on the last step
    assert float(s[:, n_obs:].std()) > 1e-3
    assert torch.equal(outs[0], outs[1]), "same seed -> bitwise identical video"


def test_window_equals_direct_p_sample_loop():
    """One window through sample_video == p_sample_loop on the same inputs and seed."""
    from improved_diffusion.video_sampler import sample_video, default_sampling_args, window_inputs
    cfg, sd, inp = load_case("micro")
    model = build_native(cfg, sd).eval()
    diff = make_diffusion(1000, "4")
    batch = torch.randn(2, 8, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    args = default_sampling_args(sampling_scheme="autoreg", n_obs=4, max_frames=8, max_latent_frames=4, device="cuda")
    torch.manual_seed(21)
    samples, used = sample_video(args, model, diff, batch, verbose=False)
    assert len(used) == 1
    start = torch.zeros_like(batch).cuda()
    start[:, :4] = batch[:, :4].cuda()
    fi, x0, om, lm = window_inputs(start, [[0, 1, 2, 3]] * 2, [[4, 5, 6, 7]] * 2)
    torch.manual_seed(21)
    direct, _ = diff.p_sample_loop(model, tuple(x0.shape), clip_denoised=True,
                                   model_kwargs=dict(frame_indices=fi, x0=x0, obs_mask=om, latent_mask=lm), latent_mask=lm)
    assert torch.equal(samples[:, 4:].cuda(), direct[:, 4:])
