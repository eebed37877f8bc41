"""Windowed long-video sampling: fill a (B, T, C, H, W) video window by window, each window one
``p_sample_loop`` over at most ``max_frames`` frames chosen by a sampling scheme.

Behaviour of ``sample_video`` in the reference's scripts/video_sample.py:28-85 (same arguments, same return
value), restructured for the MI355X path:
  * the running ``samples`` tensor lives on the device, so a window costs two device gathers and one scatter
    instead of host<->device round trips of the whole window;
  * windows of the same length share one captured hipGraph and one set of autotuned launches
    (``GaussianDiffusion._graph_sampler`` keeps the few shapes a schedule uses);
  * for latent-space models the window is written back as LATENTS (``return_decoded=False``): the reference
    writes decoded pixels into the latent tensor, which only works for pixel-space models.
"""
from types import SimpleNamespace

import torch as th

from .sampling_schemes import sampling_schemes


def window_inputs(samples, obs_frame_indices, latent_frame_indices, device=None):
    """(frame_indices, x0, obs_mask, latent_mask) of one window: observed frames first, then the latents
    (reference video_sample.py:56-61).  ``*_frame_indices`` are per-video lists."""
    B = samples.shape[0]
    device = samples.device if device is None else device
    obs = th.as_tensor(obs_frame_indices, dtype=th.long).view(B, -1)
    lat = th.as_tensor(latent_frame_indices, dtype=th.long).view(B, -1)
    frame_indices = th.cat([obs, lat], dim=1).to(device)
    rows = th.arange(B, device=samples.device)[:, None]
    x0 = samples[rows, frame_indices.to(samples.device)].to(device)      # advanced indexing copies
    obs_mask = th.cat([th.ones_like(obs), th.zeros_like(lat)], dim=1).view(B, -1, 1, 1, 1).float().to(device)
    return frame_indices, x0, obs_mask, 1 - obs_mask


@th.no_grad()
def sample_video(args, model, diffusion, batch, just_get_indices=False, verbose=True):
    """batch: (B, T, C, H, W); the first ``args.n_obs`` frames are observed, the rest are generated.

    args needs: n_obs, max_frames, max_latent_frames, sampling_scheme, clip_denoised, device and optionally
    optimality / eval_dir (an ``optimal_schedule.pt`` under eval_dir).  Returns ``(samples, indices_used)``
    with ``samples`` on batch's device and ``indices_used`` the list of (obs, latent) index lists per window.
    """
    B, T = batch.shape[:2]
    device = th.device(args.device)
    samples = th.zeros_like(batch, device=device)
    samples[:, :args.n_obs] = batch[:, :args.n_obs].to(device)
    optimality = getattr(args, "optimality", None)
    schedule_path = None if optimality is None else args.eval_dir / "optimal_schedule.pt"
    scheme = iter(sampling_schemes[args.sampling_scheme](
        video_length=T, num_obs=args.n_obs, max_frames=args.max_frames, step_size=args.max_latent_frames,
        optimal_schedule_path=schedule_path))
    batch_dev = batch.to(device) if just_get_indices else None
    decoded = getattr(diffusion, "diffusion_space", None) in (None, "pixel")
    rows = th.arange(B, device=device)[:, None]
    indices_used = []
    while True:
        scheme.set_videos(samples)         # only the adaptive schemes look at the frames
        try:
            obs_idx, lat_idx = next(scheme)
        except StopIteration:
            break
        if verbose:
            print(f"Conditioning on {sorted(obs_idx)} frames, predicting {sorted(lat_idx)}.")
        frame_indices, x0, obs_mask, latent_mask = window_inputs(samples, obs_idx, lat_idx, device)
        n_lat = len(lat_idx[0])
        if just_get_indices:
            local = batch_dev[rows, frame_indices]
        else:
            local, _ = diffusion.p_sample_loop(
                model, tuple(x0.shape), clip_denoised=args.clip_denoised,
                model_kwargs=dict(frame_indices=frame_indices, x0=x0, obs_mask=obs_mask, latent_mask=latent_mask),
                latent_mask=latent_mask, return_attn_weights=False, return_decoded=decoded)
        samples[rows, frame_indices[:, -n_lat:]] = local[:, -n_lat:]
        indices_used.append((obs_idx, lat_idx))
    return samples.to(batch.device), indices_used


def default_sampling_args(**kw):
    """Namespace with the defaults of the reference CLI (video_sample.py:171-192) for programmatic use."""
    d = dict(sampling_scheme="autoreg", n_obs=36, max_frames=20, max_latent_frames=None, clip_denoised=True,
             optimality=None, eval_dir=None, device="cuda" if th.cuda.is_available() else "cpu")
    d.update(kw)
    if d["max_latent_frames"] is None:
        d["max_latent_frames"] = d["max_frames"] // 2
    return SimpleNamespace(**d)
