#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec of the latent-video hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sample|train] [--no-cpu]

Workload (BASELINE.json configs[1]): latent U-Net num_channels=64, num_res_blocks=1, max_frames=20,
batch 2, 1000-step DDPM ancestral sampling (p_sample loop) on synthetic 4x16x16 latents.  A "step" is one
denoising step of the whole batch: timestep remap + U-Net forward + noise draw + x_{t-1} update, i.e. one
iteration of reference gaussian_diffusion.py:509-522.  Inputs/state are resident in HBM; the step is one
hipGraph replay.

N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process is one of
the ranks; started plainly as `python bench.py --gpus N` it starts the N ranks itself (a torch.distributed.run child,
before anything here touches the GPU) and passes their JSON line through.  Sampling does not communicate (the reference
parallelises sampling over videos, video_sample.py:192-200): every rank samples its own videos - weak scaling,
`value` = N*K / max-over-ranks time.  The part of the path that does exchange data - training at BASELINE.json
configs[2], batch sharded over the ranks, bucketed gradient all-reduce overlapped with the backward pass - is timed in
the same run and reported at top level for N > 1 (`train_videos_per_s`, `allreduce_bytes_per_step`,
`exposed_allreduce_ms_per_step`, `collective_world_size`, `collective_backend`) and under `train` in full.
On a box with fewer than N GPUs the ranks share the cards and the collective backend falls back to gloo (a rehearsal
of the plumbing, labelled as such in the line).

One JSON line on stdout (rank 0) with `roofline` (dominant kernel = the implicit-GEMM conv, live HIP-event timing, every
tile-shape instance listed) and `cpu_baseline` (the CPU oracle restatement of the same step: all cores, one thread, and a
training step; bounded samples).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch as th  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def make_model_and_diffusion(num_channels, device, steps=1000, respacing="", image_size=16):
    from improved_diffusion import script_util as su
    kw = su.model_and_diffusion_defaults()
    kw.update(image_size=image_size, in_channels=4, num_channels=num_channels, num_res_blocks=1, num_heads=4,
              attention_resolutions="16,8", diffusion_steps=steps, timestep_respacing=respacing,
              diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None})
    model, diffusion = su.create_model_and_diffusion(**kw)
    # random-init EVERY parameter (the reference zero-initialises some layers, which would make the
    # network output exact zeros and let the chip clock up on zero operands)
    g = th.Generator().manual_seed(7)
    with th.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * th.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(th.randn(p.shape, generator=g) / math.sqrt(fan_in))
    return model.to(device).eval(), diffusion


def synthetic_inputs(B, T, rank, device):
    g = th.Generator().manual_seed(1234 + rank)
    x0 = th.randn(B, T, 4, 16, 16, generator=g)
    fi = th.stack([th.arange(T) if b % 2 == 0 else th.sort(th.randperm(1000, generator=g)[:T])[0] for b in range(B)])
    obs = th.zeros(B, T, 1, 1, 1)
    obs[:, :T // 3] = 1.0
    kw = dict(x0=x0, frame_indices=fi, obs_mask=obs, latent_mask=1.0 - obs)
    return {k: v.to(device) for k, v in kw.items()}


def conv_flops(a):
    M = a.N * a.Ho * a.Wo
    K = a.ksize * a.ksize * (a.C0 + a.C1) + a.s2C0 + a.s2C1
    return 2.0 * M * a.Cout * K


def conv_bytes(a):
    """Compulsory bytes of one implicit-GEMM launch: every operand once (sources, packed weights, residual) + the output."""
    M = a.N * a.Ho * a.Wo
    src = a.N * a.Hs * a.Ws * (a.C0 + a.C1) + M * (a.s2C0 + a.s2C1)
    w = a.Cout * (a.ksize * a.ksize * (a.C0 + a.C1) + a.s2C0 + a.s2C1)
    return 4.0 * (src + w + M * a.Cout * (2 if a.res else 1))


def kernel_breakdown(plan, reps=10, inner=4):
    """Per-launch HIP-event timing of every step of the forward plan (eager, on the stream the kernels are launched
    on).  Each step is launched `inner` times back to back between one event pair, which amortises the ~5 us that
    an event pair adds around a single launch, so the figure tracks the kernel duration rocprofv3 reports."""
    import ctypes as C
    from improved_diffusion import _native as nat
    L = nat.lib()
    s = nat.stream()
    n = len(plan.steps)
    obs = [[] for _ in range(n)]
    ev = [(th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) for _ in range(n)]
    for rep in range(reps + 2):
        for i, (fn, args) in enumerate(plan.steps):
            ev[i][0].record()
            for _ in range(inner):
                fn(*args, s)
            ev[i][1].record()
        th.cuda.synchronize()
        if rep >= 2:
            for i in range(n):
                obs[i].append(ev[i][0].elapsed_time(ev[i][1]) / inner)
    # median over the repetitions (x reps, so that the sums below read as before): one stalled repetition - a 40 ms hiccup of
    # one launch behind a 97-window video was seen once - must not become that kernel's "duration"
    tot = [sorted(o)[len(o) // 2] * reps for o in obs]
    groups = {}
    for i, (fn, args) in enumerate(plan.steps):
        name = fn.__name__
        flops = nbytes = 0.0
        if name == "lfvdm_conv_igemm":
            a = args[0]._obj
            nt, nw = C.c_int(), C.c_int()
            L.lfvdm_conv_igemm_config(C.byref(a), C.byref(nt), C.byref(nw))
            v = nt.value
            name = f"conv_igemm_kernel<{v // 1000},{v // 100 % 10},{v // 10 % 10},{v % 10}>"
            flops = conv_flops(a)
            nbytes = conv_bytes(a)
        elif name == "lfvdm_level_chain":
            # persistent level chain: the implicit-GEMM body (+ small GroupNorms) of several stages in ONE launch; FLOPs and
            # compulsory bytes are those of the stand-alone launches it replaces
            name = "level_chain_kernel"
            ch = next(c for c in plan.chains if c["step"][1] is args)
            for f2, a2 in ch["steps"]:
                if f2.__name__ == "lfvdm_conv_igemm":
                    flops += conv_flops(a2[0]._obj)
                    nbytes += conv_bytes(a2[0]._obj)
                elif f2.__name__ == "lfvdm_gn_apply_part":       # (src, C, N, P, ...): read once, written once
                    nbytes += 2.0 * 4.0 * a2[1] * a2[2] * a2[3]
                else:                                            # lfvdm_gn_apply (src0, src1, C0, C1, N, P, ...)
                    nbytes += 2.0 * 4.0 * a2[4] * a2[5] * (a2[2] + a2[3])
        elif name in ("lfvdm_gn_apply", "lfvdm_gn_apply_ws"):      # (src0, src1, C0, C1, N, P, ...): read once, written once
            nbytes = 2.0 * 4.0 * args[4] * args[5] * (args[2] + args[3])
        elif name == "lfvdm_gn_temporal":                           # (x, gamma, beta, eps, y, B, T, P, C)
            nbytes = 2.0 * 4.0 * args[5] * args[6] * args[7] * args[8]
        elif name == "lfvdm_proj_gn":                               # (o, W, b, res, gamma, beta, eps, act, out, raw, N, P, C)
            rows, Cc = args[10] * args[11], args[12]
            flops = 2.0 * rows * Cc * Cc
            nbytes = 4.0 * (rows * Cc * (3 + (args[9] is not None)) + Cc * Cc)   # o, residual read; normalised (+ raw) sum written
        elif name == "lfvdm_gn_temporal_qkv":                       # (x, gamma, beta, eps, xn, W, b, qkv, B, T, P, C)
            rows, Cc = args[8] * args[9] * args[10], args[11]
            flops = 2.0 * rows * 3 * Cc * Cc
            nbytes = 4.0 * (rows * Cc * 2 + rows * 3 * Cc + 3 * Cc * Cc)       # x read, xn + qkv written, the filters once
        gsum = groups.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        gsum["launches"] += 1
        gsum["ms"] += tot[i] / reps
        gsum["flops"] += flops
        gsum["bytes"] += nbytes
    return groups


def _host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))  # the GPU box grants 16 host cores per GPU


def _oracle_model(model):
    from oracle import unet_oracle as uo
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = uo.make_cfg(model_channels=model.model_channels, channel_mult=model.channel_mult,
                      attention_resolutions=model.attention_resolutions, num_heads=model.num_heads)
    return sd, cfg


def cpu_baseline_sample(model, inputs, B, T, threads, budget_s, H=16, chain=1000, what="the same workload"):
    """CPU oracle restatement of the same denoising step (kind 'port') on `threads` host threads."""
    from oracle import unet_oracle as uo, diffusion_oracle as do
    th.set_num_threads(threads)
    sd, cfg = _oracle_model(model)
    tab = do.Tables(do.linear_betas(chain))
    ci = {k: v.cpu() for k, v in inputs.items()}
    x = th.randn(B, T, 4, H, H)
    n, t0 = 0, None
    with th.no_grad():
        for i in list(range(chain - 1, -1, -1)) * (1 + 40 // chain):
            if n == 2:
                t0 = time.perf_counter()
            t = th.full((B,), i, dtype=th.long)
            eps, _ = uo.unet_forward(sd, cfg, x, ci["x0"], do.model_timesteps(tab, t), ci["frame_indices"],
                                     ci["obs_mask"], ci["latent_mask"])
            x, _ = do.p_sample(tab, eps, x, t, th.randn_like(x))
            n += 1
            if t0 is not None and time.perf_counter() - t0 > budget_s:
                break
    el = time.perf_counter() - t0
    return dict(value=round((n - 2) / el, 3), unit="steps/s", cores=threads, kind="port",
                sample=f"{n - 2} p_sample steps of {what} (oracle/unet_oracle.py + diffusion_oracle.py, torch CPU fp32)")


def cpu_baseline_train(threads, budget_s, num_channels=128, B=2, T=20, H=16, chain=1000,
                       what="configs[2] (ch128, batch 2, 20 frames)"):
    """CPU oracle restatement of one optimizer step (default: BASELINE.json configs[2]; reference TrainLoop.run_step,
    train_util.py:267-275: zero grads, q_sample, U-Net forward, masked MSE, backward, AdamW, EMA) on `threads` threads."""
    from oracle import unet_oracle as uo, diffusion_oracle as do
    th.set_num_threads(threads)
    model, _ = make_model_and_diffusion(num_channels, th.device("cpu"), steps=chain, image_size=H)
    sd, cfg = _oracle_model(model)
    del model
    params = {k: v.requires_grad_(True) for k, v in sd.items()}
    ema = {k: v.detach().clone() for k, v in params.items()}
    opt = th.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=0.0)
    tab = do.Tables(do.linear_betas(chain))
    g = th.Generator().manual_seed(4321)
    obs = th.zeros(B, T, 1, 1, 1); obs[:, :max(T // 3, 1)] = 1.0
    lat = 1.0 - obs
    if T >= 10:
        lat[:, -3:] = 0.0                                   # 3 padding frames: neither observed nor latent
    n, t0 = 0, None
    while True:
        if n == 1:
            t0 = time.perf_counter()
        x0 = th.randn(B, T, 4, H, H, generator=g).clamp(-1, 1)
        fi = th.stack([th.sort(th.randperm(40, generator=g)[:T])[0] for _ in range(B)])
        t = th.randint(0, chain, (B,), generator=g)
        noise = th.randn(x0.shape, generator=g)
        opt.zero_grad(set_to_none=True)

        def model_fn(x_t, ts):
            return uo.unet_forward(params, cfg, x_t, x0, ts, fi, obs, lat)[0]

        losses = do.training_losses(tab, model_fn, x0, t, noise, 1 - obs, lat)
        losses["loss"].mean().backward()
        opt.step()
        with th.no_grad():
            for k, v in params.items():
                ema[k].mul_(0.9999).add_(v.detach(), alpha=1 - 0.9999)
        n += 1
        if t0 is not None and time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return dict(value=round((n - 1) / el, 3), unit="optimizer steps/s", cores=threads, kind="port",
                sample=f"{n - 1} training steps at {what}: oracle forward + torch autograd "
                       "backward + torch AdamW + EMA, torch CPU fp32")


def train_step_flops(model, B, T, H, W):
    """Algorithmic FLOPs of ONE training step at this shape: 3 x the forward (forward + data gradient + weight gradient
    of every contraction; SURVEY.md §8d "train step = 3x") with the forward counted in closed form - implicit GEMMs
    2*M*Cout*K from the forward plan's launch list, temporal attention 10*B*P*T^2*C, spatial attention 4*N*P^2*C."""
    from improved_diffusion import _native as nat
    from improved_diffusion._engine import Plan
    pl = Plan(model.native_engine(), B, T, H, W, False)
    L = nat.lib()
    conv = sum(conv_flops(a[0]._obj) for fn, a in pl.steps if fn is L.lfvdm_conv_igemm)
    att = 0.0
    for fn, a in pl.steps:
        if fn is L.lfvdm_attn_temporal or fn is L.lfvdm_attn_temporal_sel or fn is L.lfvdm_attn_temporal_ring:
            Bv, Tv, P, C = a[7], a[8], a[9], a[10]
            att += 10.0 * Bv * P * Tv * Tv * C
        elif fn is L.lfvdm_attn_spatial:
            N, P, C = a[4], a[5], a[6]
            att += 4.0 * N * P * P * C
        elif fn is L.lfvdm_attn_spatial_fused:     # qkv projection + core in one launch: count both
            N, P, C = a[4], a[5], a[6]
            conv += 2.0 * N * P * 3 * C * C
            att += 4.0 * N * P * P * C
        elif fn is L.lfvdm_gn_temporal_qkv:        # (x, gamma, beta, eps, xn, W, b, qkv, B, T, P, C): the temporal qkv projection
            Bv, Tv, P, C = a[8], a[9], a[10], a[11]
            conv += 2.0 * Bv * Tv * P * 3 * C * C
        elif fn is L.lfvdm_proj_gn:                # (o, W, b, res, gamma, beta, eps, act, out, raw, N, P, C): an attention output projection
            N, P, C = a[10], a[11], a[12]
            conv += 2.0 * N * P * C * C
    return {"forward_conv_gemm": conv, "forward_attention": att, "step": 3.0 * (conv + att)}


_TRAIN_FAMILIES = (
    ("implicit GEMM forward + data gradient", ("conv_igemm_kernel",)),
    ("weight gradients", ("conv_wgrad", "unpack_conv_grad", "wgrad")),
    ("attention forward + backward", ("attn_", "rpe_")),
    ("GroupNorm forward + backward", ("gn_",)),
    ("optimizer (AdamW + EMA)", ("adamw_ema",)),
    ("embedding network (row-dot)", ("rowdot", "silu_kernel")),
)


def running_code_stamp():
    """What identifies the code that is running: the library's ABI version and the digest of csrc/* + headers recorded by
    the build next to the library (lib/build_manifest.json)."""
    from improved_diffusion import _native as nat
    digest, origin = None, None
    pkg = os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd")
    try:
        with open(os.path.join(pkg, "lib", "build_manifest.json")) as f:
            digest, origin = json.load(f).get("source_digest"), "build manifest"
    except (OSError, ValueError):
        pass
    if digest is None:          # library shipped without its manifest: the digest of the sources next to it
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("lfvdm_build", os.path.join(pkg, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            digest, origin = mod.source_digest(), "source tree"
        except Exception:
            pass
    return {"abi": int(nat.lib().lfvdm_abi_version()), "source_digest": digest, "digest_from": origin}


def profile_stamp(path):
    """Stamp of a committed profile: `<file>.stamp.json` written next to it by tools/refresh_profiles.sh on the box that took
    it ({"abi", "source_digest", "git_head"}).  -> (stamp or None, stale?) where stale means it was taken with other kernels
    than the ones running now (or carries no stamp at all)."""
    try:
        with open(path + ".stamp.json") as f:
            st = json.load(f)
    except (OSError, ValueError):
        return None, True
    now = running_code_stamp()
    stale = st.get("abi") != now["abi"] or (now["source_digest"] is not None and st.get("source_digest") != now["source_digest"])
    return st, bool(stale)


def train_family_split(pattern="r*_train_kernel_stats.csv"):
    """Per-family GPU milliseconds of one training step from the newest committed rocprofv3 kernel summary
    (profiles/rNN_train_kernel_stats.csv: `rocprofv3 --kernel-trace --stats -- python3 tools/train_profile.py`; steps =
    calls of the fused optimizer kernel).  None when no summary is committed.  `stale: true` when the summary's stamp
    (library ABI + source digest of the run that took it) is not the running code's."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True)
    if not paths:
        return None
    rows = []
    with open(paths[0]) as f:
        for r in csv.DictReader(f):
            rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
    steps = sum(c for n, c, _, _ in rows if "adamw_ema" in n)
    if steps <= 0:
        return None
    fam = {k: 0.0 for k, _ in _TRAIN_FAMILIES}
    fam["other kernels"] = 0.0
    launches = 0
    for n, c, ns, _ in rows:
        launches += c
        for k, pats in _TRAIN_FAMILIES:
            if any(pt in n for pt in pats):
                fam[k] += ns
                break
        else:
            fam["other kernels"] += ns
    gn = [(n, avg) for n, _, _, avg in rows if "gn_" in n]
    stamp, stale = profile_stamp(paths[0])
    return {"source": os.path.relpath(paths[0], ROOT), "stamp": stamp, "stale": stale,
            "profiled_steps": steps, "launches_per_step": round(launches / steps, 1),
            "ms_per_step": {k: round(v / steps / 1e6, 3) for k, v in fam.items()},
            "gpu_ms_per_step": round(sum(fam.values()) / steps / 1e6, 3),
            "slowest_groupnorm_kernel_avg_us": round(max((a for _, a in gn), default=0.0) / 1e3, 1)}


def synthetic_video_stream(B, T_video, seed):
    g = th.Generator().manual_seed(seed)
    while True:
        yield (th.randn(B, T_video, 4, 16, 16, generator=g).clamp(-1, 1), {})


def _event_time_us(fn, reps=20, warm=3):
    """Average HIP-event time of `fn` (which enqueues on torch's current stream = the stream the C-ABI launches use)."""
    for _ in range(warm):
        fn()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    th.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


def _hbm_record(nbytes, us, what):
    gbs = nbytes / (us * 1e-6) / 1e9
    return {"bytes": int(nbytes), "us": round(us, 2), "gb_per_s": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "frac": round(gbs / HBM_PEAK_GBS, 4), "what": what}


def hbm_phases_latent(diffusion, dev):
    """The memory-bound phases of the path against the HBM peak, live HIP-event timing of the same entry
    points the training step uses, at the configs[2] shapes (B=2, T=20, 4x16x16): q_sample (3 tensors) and masked MSE
    (eps, eps_hat read; per-sample sums written).  These tensors are 164 KB: the launches are latency-bound and the
    fraction says so."""
    from improved_diffusion._autograd import masked_mse
    B, T = 2, 20
    g = th.Generator(device="cpu").manual_seed(5)
    x0, noise = th.randn(B, T, 4, 16, 16, generator=g).to(dev), th.randn(B, T, 4, 16, 16, generator=g).to(dev)
    t = th.tensor([17, 801], device=dev)
    mask = th.ones(B, T, 1, 1, 1, device=dev)
    n = x0.numel()
    with th.no_grad():
        q = _event_time_us(lambda: diffusion.q_sample(x0, t, noise))
        m = _event_time_us(lambda: masked_mse(noise, x0, mask))
    return {"q_sample": _hbm_record(12.0 * n, q, "x0, eps read + x_t written, 40960 elements (incl. the output allocation)"),
            "masked_mse": _hbm_record(8.0 * n, m, "eps, eps_hat read, 40960 elements")}


def hbm_phases_pixel(dev):
    """Large-map GroupNorm forward / backward (the chunked two-launch forms) on one pixel-space activation,
    20 x 128 x 128 x 128 fp32 = 168 MB: forward reads x twice and writes act once (3 tensors), backward reads x and da
    twice and writes dx once (5 tensors)."""
    from improved_diffusion import _backward as bw, _native as nat
    N, P, C = 20, 128 * 128, 128
    g = th.Generator(device="cpu").manual_seed(6)
    x = th.randn(N * P // 64, C, generator=g).to(dev).repeat(64, 1)
    da = th.randn(N * P // 64, C, generator=g).to(dev).repeat(64, 1)
    gamma, beta = th.nn.Parameter(th.ones(C, device=dev)), th.nn.Parameter(th.zeros(C, device=dev))
    res = {}
    with th.no_grad():
        _, cA, cB, st = bw._gn_apply(x, None, C, 0, N, P, gamma, beta, None, 1, nat.ACT_SILU)
        f = _event_time_us(lambda: bw._gn_apply(x, None, C, 0, N, P, gamma, beta, None, 1, nat.ACT_SILU), reps=10)
        b = _event_time_us(lambda: bw._gn_backward(da, x, None, C, 0, N, P, cA, cB, st, nat.ACT_SILU, gamma, beta, None, 1,
                                                   inplace=True), reps=10)
    nb = 4.0 * N * P * C
    res["gn_apply_large_map"] = _hbm_record(3 * nb, f, "lfvdm_gn_apply_ws (2 launches) on 20x128x128x128: x read twice, act written")
    res["gn_backward_large_map"] = _hbm_record(5 * nb, b, "lfvdm_gn_bwd_ws (2 launches) on 20x128x128x128: x, da read twice, dx written")
    return res


def exchange_machinery_probe(dev, steps=12):
    """What the bucketed exchange costs BEFORE any byte crosses xGMI, on today's code: the cfg-C training step once more
    at world size 1 with LFVDM_FORCE_EXCHANGE=1 - marker nodes, per-bucket folds, counter kernels, one SUM all-reduce per
    bucket over ONE rank on RCCL's stream behind the polling kernel, the collective skip word, the optimizer behind them."""
    os.environ["LFVDM_FORCE_EXCHANGE"] = "1"
    try:
        # (8 untimed steps: the first replays next to a freshly created communicator run 0.2-0.3 ms slow - round 5 measured
        # 7.64 ms over steps 5-16 and 7.38 over steps 6-25 of the same job, tools/exchange_probe.py)
        probe = bench_train(0, 1, dev, steps, 8, probe_only=True)
        # The same step with ONE bucket behind the graph's end and no device-side waits: what is left is the collective
        # itself - at world size 1 RCCL's "all-reduce" is a device copy of the 122 MB arena - i.e. the share of the figure
        # above that is RCCL's, not the machinery's (markers, folds, counters, polling kernels, five launches)
        saved = {k: os.environ.get(k) for k in ("LFVDM_GRAD_BUCKETS", "LFVDM_OVERLAP_EXCHANGE")}
        os.environ.update(LFVDM_GRAD_BUCKETS="1", LFVDM_OVERLAP_EXCHANGE="0")
        try:
            probe["one_bucket_behind_the_graph_ms_per_step"] = bench_train(0, 1, dev, steps, 8, probe_only=True)["ms_per_step"]
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return probe
    finally:
        os.environ.pop("LFVDM_FORCE_EXCHANGE", None)


def bench_train(rank, world, dev, steps, warmup, probe_only=False):
    """DDP-style training at BASELINE.json configs[2]: latent U-Net num_channels=128, max_frames=20, batch 2 per
    GPU (global 2*N), one optimizer step = mask sampling + q_sample + U-Net forward/backward + ONE RCCL all-reduce
    of the gradient arena + fused AdamW/EMA (reference TrainLoop.run_step, train_util.py:267-275)."""
    import argparse as ap
    from improved_diffusion.train_util import TrainLoop
    model, diffusion = make_model_and_diffusion(128, dev)
    model.train()
    loop = TrainLoop(model=model, diffusion=diffusion, data=synthetic_video_stream(2, 40, 4321 + rank), batch_size=2,
                     microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                     resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                     schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                     pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
    th.manual_seed(99 + rank)
    np.random.seed(99 + rank)
    for _ in range(warmup):
        loop.run_step()
        loop.step += 1
    th.cuda.synchronize()
    if world > 1:
        dist.barrier()
    th.cuda.synchronize()
    loop.host_wait_s = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.run_step()
        loop.step += 1
    host_wall = time.perf_counter() - t0     # host side of the loop: work + waiting for the GPU (it runs one step behind)
    host_wait = loop.host_wait_s             # ... of which blocked on GPU events (staging-slot reuse, deferred loss log)
    host = host_wall - host_wait
    th.cuda.synchronize()
    if world > 1:
        dist.barrier()
    th.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = th.tensor([el], device=dev, dtype=th.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    from improved_diffusion.logger import logger
    loop._flush_loss_log()
    loss = float(logger.name2val.get("loss", float("nan")))
    logger.dumpkvs()
    P = sum(p.numel() for p in model.parameters())
    xch = loop.exchange
    th.cuda.synchronize()
    if probe_only:
        xch.collect_timing()
        return {"ms_per_step": round(1000.0 * el / steps, 3), "steps": steps, "overlap_probe": xch.overlap_probe,
                "overlap_with_backward": bool(xch.overlap), "buckets": len(xch.ranges),
                "buckets_started_inside_the_backward": xch.stats["buckets_behind_event"],
                "buckets_started_after_the_backward": xch.stats["buckets_behind_graph_end"],
                "exposed_ms_per_step": round(sum(xch.exposed_ms[-steps:]) / max(1, len(xch.exposed_ms[-steps:])), 3),
                "backend": dist.get_backend() if dist.is_initialized() else None}
    exposed = xch.collect_timing()[-steps:] if world > 1 else []
    opt_us = None
    if world == 1:          # the fused optimizer launch alone (HBM-bound: 5 arenas read, 4 written)
        def _opt():
            loop.optimize_normal()
        opt_us = _event_time_us(_opt, reps=10, warm=2)
    fl = train_step_flops(model, 2, 20, 16, 16)
    tfl = fl["step"] / (el / steps) / 1e12
    roof = {"bound": "mfma", "flops_per_step": fl["step"], "forward_conv_gemm_flops": fl["forward_conv_gemm"],
            "forward_attention_flops": fl["forward_attention"], "achieved": round(tfl, 2), "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
            "note": "whole optimizer step (wall clock, all launches) against the fp32 MFMA peak; FLOPs = 3 x forward "
                    "(implicit GEMMs 2*M*Cout*K from the launch list + attention in closed form)",
            "families": train_family_split()}
    out = {"optimizer_steps_per_s": round(steps / el, 3), "ms_per_step": round(1000.0 * el / steps, 2), "steps": steps,
           "roofline": roof,
           "host_issue_ms_per_step": round(1000.0 * host / steps, 2),
           "host_wait_ms_per_step": round(1000.0 * host_wait / steps, 2),
           "global_batch": 2 * world, "videos_per_s": round(2 * world * steps / el, 2), "params": P,
           "allreduce_bytes_per_step": 4 * loop.arena.numel if world > 1 else 0, "last_loss": loss,
           "workload": "train: U-Net num_channels=128 num_res_blocks=1 max_frames=20 batch 2/GPU, AdamW+EMA, bucketed "
                       "all-reduce of the fp32 gradient arena overlapped with the backward graph (BASELINE.json configs[2])"}
    if opt_us is not None:
        out["hbm_phases"] = {"adamw_ema": _hbm_record(4.0 * P * (7 + 2 * len(loop.ema_flat)), opt_us,
                                                      "lfvdm_adamw_ema: p, g, m, v, ema read; p, m, v, ema written = 36 B per "
                                                      "parameter (+ the 4-byte grad-norm memset of optimize_normal)")}
    # the same record at every N (N = 1: no collective runs, the layout is what the N > 1 job will exchange)
    out["exchange"] = {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": dist.get_backend() if (dist.is_initialized() and world > 1) else "none (one rank)",
                       "buckets": len(xch.ranges), "bucket_bytes": [4 * (hi - lo) for lo, hi in xch.ranges],
                       "overlap_with_backward": bool(xch.overlap) if world > 1 else None,
                       "overlap_probe": xch.overlap_probe,
                       "exposed_ms_per_step": (round(sum(exposed) / max(1, len(exposed)), 3) if exposed else None) if world > 1 else 0.0,
                       "buckets_started_inside_the_backward": xch.stats["buckets_behind_event"],
                       "buckets_started_after_the_backward": xch.stats["buckets_behind_graph_end"]}
    return out


def bench_long_video(dev, max_windows, batch=1):
    """BASELINE.json configs[3]: 1000-frame videos generated by the hierarchy-2 schedule in windows of <= 20 frames, 250
    respaced steps per window (97 windows with 36 observed frames), cfg-B network.  batch = videos generated side by side
    (1: the configuration as BASELINE.json words it; 8: the reference's default, scripts/video_sample.py:171 --batch_size=8,
    :88-99 - eight videos share every window's index lists)."""
    from types import SimpleNamespace
    from improved_diffusion.video_sampler import sample_video, default_sampling_args
    from improved_diffusion.sampling_schemes import sampling_schemes
    model, diffusion = make_model_and_diffusion(64, dev, respacing="250")
    T = 1000
    g = th.Generator().manual_seed(77)
    video = (th.randn(batch, 1, 4, 16, 16, generator=g) + 0.1 * th.randn(batch, T, 4, 16, 16, generator=g).cumsum(1)) * 0.5
    args = default_sampling_args(sampling_scheme="hierarchy-2", n_obs=36, max_frames=20, max_latent_frames=10, device=str(dev))
    if max_windows < 97:      # truncated run: shorten the video so that the schedule ends early
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            while True:
                n = sum(1 for _ in iter(sampling_schemes["hierarchy-2"](video_length=T, num_obs=36, max_frames=20, step_size=10)))
                if n <= max_windows or T <= 60:
                    break
                T -= 10
        video = video[:, :T].contiguous()
    th.manual_seed(5)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        samples, used = sample_video(args, model, diffusion, video, verbose=False)
    th.cuda.synchronize()
    el = time.perf_counter() - t0
    steps = len(used) * diffusion.num_timesteps
    lengths = sorted({len(o[0]) + len(l[0]) for o, l in used})
    rec = {"workload": f"long video: hierarchy-2, T={T}, K=20, step=10, n_obs=36, respacing 250, batch {batch} (BASELINE.json "
                       "configs[3]" + ("" if batch == 1 else "; the reference's default --batch_size") + ")",
           "batch": batch, "windows": len(used), "window_lengths": lengths, "denoising_steps": steps, "seconds": round(el, 2),
           "steps_per_s_incl_setup": round(steps / el, 1), "frames_generated_per_s": round(batch * (T - 36) / el, 2),
           "frame_steps_per_s": round(batch * 20 * steps / el, 1), "finite": bool(th.isfinite(samples).all())}
    # the implicit-GEMM body at this batch's launch shapes (M = batch x 20 x H x W rows): live event timing of the 20-frame
    # window's plan, as for the headline workload
    try:
        sm = next(v for k, v in diffusion._samplers.items() if k[1][1] == 20)
        groups = kernel_breakdown(sm.plan, reps=3, inner=2)
        convs = {k: v for k, v in groups.items() if k.startswith("conv_igemm") or k == "level_chain_kernel"}
        fl, ms = sum(v["flops"] for v in convs.values()), sum(v["ms"] for v in convs.values())
        rec.update({"launches_per_step": len(sm.plan.steps), "conv_gemm_gflop_per_step": round(fl / 1e9, 2),
                    "conv_gemm_us_per_step": round(1000 * ms, 1), "conv_gemm_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                    "conv_gemm_frac_of_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                    "level_chains": [dict(stages=c["n"], work_items=c["items"]) for c in sm.plan.chains],
                    "time_table_fallback": sm.plan.time_table_fallback})
    except StopIteration:
        pass
    return rec


def bench_pixel(dev, steps):
    """BASELINE.json configs[4]: pixel-space stress, 128x128x3 frames, max_frames=20, num_channels=128 (reference
    defaults: num_res_blocks=2, channel_mult (1,1,2,3,4), attention at 16x16 and 8x8), batch 1; denoising steps of
    the captured sampler plus the conv roofline of that step."""
    from improved_diffusion import script_util as su
    kw = su.model_and_diffusion_defaults()
    kw.update(image_size=128, in_channels=3, num_channels=128, num_res_blocks=2, num_heads=4, attention_resolutions="16,8",
              diffusion_steps=1000, timestep_respacing="",
              diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None})
    model, diffusion = su.create_model_and_diffusion(**kw)
    g = th.Generator().manual_seed(7)
    with th.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * th.randn(p.shape, generator=g))
            else:
                p.copy_(th.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))
    model = model.to(dev).eval()
    B, T = 1, 20
    shape = (B, T, 3, 128, 128)
    g = th.Generator().manual_seed(11)
    obs = th.zeros(B, T, 1, 1, 1)
    obs[:, :T // 3] = 1.0
    inputs = {k: v.to(dev) for k, v in dict(x0=th.randn(*shape, generator=g).clamp(-1, 1), frame_indices=th.arange(T)[None],
                                            obs_mask=obs, latent_mask=1.0 - obs).items()}
    sampler = diffusion._graph_sampler(model, shape, True)
    th.manual_seed(3)
    sampler.begin(th.randn(*shape, device=dev), inputs)
    for _ in range(2):
        sampler.step(sampler.expected_t)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sampler.step(sampler.expected_t)
    th.cuda.synchronize()
    el = time.perf_counter() - t0
    groups = kernel_breakdown(sampler.plan, reps=2, inner=1)
    convs = {k: v for k, v in groups.items() if k.startswith("conv_igemm")}
    fl, ms = sum(v["flops"] for v in convs.values()), sum(v["ms"] for v in convs.values())
    return {"workload": "pixel space 128x128x3, 20 frames, batch 1, num_channels=128, num_res_blocks=2 (BASELINE.json configs[4])",
            "steps": steps, "ms_per_step": round(1000 * el / steps, 2), "steps_per_s": round(steps / el, 2),
            "params": sum(p.numel() for p in model.parameters()), "conv_gemm_gflop_per_step": round(fl / 1e9, 1),
            "conv_gemm_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "conv_gemm_frac_of_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
            "whole_step_tflops": round(fl / (el / steps) / 1e12, 1), "finite": bool(th.isfinite(sampler.plan.x_in).all())}


def make_pixel_model(dev, num_res_blocks=2):
    """BASELINE.json configs[4] network: pixel space 128x128x3, num_channels=128, reference defaults channel_mult
    (1,1,2,3,4), attention at 16x16 and 8x8; random-init every parameter (see make_model_and_diffusion)."""
    from improved_diffusion import script_util as su
    kw = su.model_and_diffusion_defaults()
    kw.update(image_size=128, in_channels=3, num_channels=128, num_res_blocks=num_res_blocks, num_heads=4,
              attention_resolutions="16,8", diffusion_steps=1000, timestep_respacing="",
              diffusion_space_kwargs={"diffusion_space": "pixel", "pre_encoded": False, "pre_encoded_stats_dict": None})
    model, diffusion = su.create_model_and_diffusion(**kw)
    g = th.Generator().manual_seed(7)
    with th.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * th.randn(p.shape, generator=g))
            else:
                p.copy_(th.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))
    return model.to(dev), diffusion


def _repeat_batch(batch):
    while True:
        yield (batch, {})


def bench_pixel_train(dev, steps, batch=1, num_res_blocks=1, warmup=4):
    """Pixel-space TRAINING, the reference's published recipe (README.md:54-57: video_train.py --batch_size=2 --max_frames 20
    --dataset=carla_no_traffic --num_res_blocks=1, i.e. 128x128x3 frames, num_channels=128; BASELINE.json configs[4] in
    video_train.py flags): one TrainLoop optimizer step = batch preparation + q_sample + U-Net forward + masked MSE +
    backward + fused AdamW/EMA (train_util.py:277-357).  FLOPs = 3 x forward, against the fp32 MFMA peak over the WHOLE
    step (wall clock)."""
    import argparse as ap
    from improved_diffusion.train_util import TrainLoop
    model, diffusion = make_pixel_model(dev, num_res_blocks)
    model.train()
    T = 20
    g = th.Generator().manual_seed(4321)
    video = th.randn(batch, 24, 3, 128, 128, generator=g).clamp(-1, 1)
    loop = TrainLoop(model=model, diffusion=diffusion, data=_repeat_batch(video), batch_size=batch, microbatch=-1, lr=1e-4,
                     ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9, resume_checkpoint="", use_fp16=False,
                     diffusion_space_kwargs={}, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.0,
                     lr_anneal_steps=0, sample_interval=None, pad_with_random_frames=True, max_frames=T,
                     enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
    th.manual_seed(99)
    np.random.seed(99)
    for _ in range(warmup):
        loop.run_step()
        loop.step += 1
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.run_step()
        loop.step += 1
    th.cuda.synchronize()
    el = time.perf_counter() - t0
    from improved_diffusion.logger import logger
    loop._flush_loss_log()
    loss = float(logger.name2val.get("loss", float("nan")))
    logger.dumpkvs()
    fl = train_step_flops(model, batch, T, 128, 128)
    tfl = fl["step"] / (el / steps) / 1e12
    P = sum(p.numel() for p in model.parameters())
    peak_gb = th.cuda.max_memory_allocated(dev) / 2 ** 30
    out = {"workload": f"pixel-space training: 128x128x3, 20 frames, batch {batch}, num_channels=128, num_res_blocks={num_res_blocks} "
                       "(reference README.md:54-57 recipe / BASELINE.json configs[4])",
           "steps": steps, "ms_per_step": round(1000.0 * el / steps, 2), "optimizer_steps_per_s": round(steps / el, 3),
           "frames_per_s": round(batch * T * steps / el, 1), "params": P, "last_loss": loss,
           "graph_replay": loop._graph_state.get("graph") is not None, "hbm_allocated_peak_gib": round(peak_gb, 2),
           "roofline": {"bound": "mfma", "flops_per_step": fl["step"], "forward_conv_gemm_flops": fl["forward_conv_gemm"],
                        "forward_attention_flops": fl["forward_attention"], "achieved": round(tfl, 2),
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
                        "note": "whole optimizer step (wall clock, all launches) against the fp32 MFMA peak; FLOPs = 3 x forward"}}
    if batch == 1 and num_res_blocks == 1:
        out["roofline"]["families"] = train_family_split("r*_pixel_train_kernel_stats.csv")
    del loop, model
    th.cuda.empty_cache()
    return out


def pmc_traffic_source():
    """(path, stamp, stale) of the newest committed PMC summary."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
    if not paths:
        return None
    stamp, stale = profile_stamp(paths[0])
    return {"source": os.path.relpath(paths[0], ROOT), "stamp": stamp, "stale": stale}


def pmc_traffic(kernel_name):
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 counter passes (newest
    profiles/rNN_pmc_traffic.json, produced by tools/pmc_target.py + tools/pmc_summarize.py); None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                v = json.load(f)["kernels"].get(kernel_name, {}).get("hbm_bytes_per_launch")
            if v is not None:
                return v
        except (OSError, ValueError, KeyError):
            continue
    return None


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, the way the
    driver does) and exit with their code.  Nothing in THIS process has touched the GPU yet (device_count() does not
    initialise it); the ranks are ordinary children, not an exec of this process."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def timed_sampling(sampler, steps, world, dev, min_seconds, n_t):
    """Regions of EXACTLY `steps` denoising steps, each bracketed by barrier + synchronize on both sides and reduced with
    MAX over the ranks; repeated until at least `min_seconds` have been timed (a 20-step region is 26 ms: too short to
    quote a rate from).  Every region walks the chain from its top (t = n_t - 1 downwards, restarting if `steps` exceeds the
    chain), so that the in-stream refills of the rolling R-table window are inside the timed region in the proportion a real
    chain has them.  -> list of region times."""
    times = []
    while True:
        th.cuda.synchronize()
        if world > 1:
            dist.barrier()
        th.cuda.synchronize()
        t0 = time.perf_counter()
        left = steps
        while left > 0:
            n = min(left, n_t)
            sampler.run(n_t - 1, n)      # = n x sampler.step, 8 steps per graph launch (GraphSampler.run)
            left -= n
        th.cuda.synchronize()
        if world > 1:
            dist.barrier()
        th.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:           # the slowest rank's time; every rank then takes the same decision to go on
            tt = th.tensor([el], device=dev, dtype=th.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        times.append(el)
        if sum(times) >= min_seconds or len(times) >= 200:
            return times


def main():
    # tile shapes measured once on an MI355X and committed (loaded read-only): the same kernels run in every bench /
    # profile pass
    os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step timed region until this much is timed")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--train-steps", type=int, default=30, help="timed optimizer steps of the training leg (0 = skip)")
    ap.add_argument("--machinery-steps", type=int, default=20,
                    help="N = 1 only: optimizer steps of the forced-exchange probe (LFVDM_FORCE_EXCHANGE=1; 0 = skip)")
    ap.add_argument("--pixel-train-steps", type=int, default=5, help="timed optimizer steps of each pixel-space training leg (0 = skip)")
    ap.add_argument("--pixel-steps", type=int, default=5, help="timed steps of the pixel-space stress config, configs[4] (0 = skip)")
    ap.add_argument("--long-video-windows", type=int, default=4,
                    help="windows of the hierarchy-2 long-video leg, configs[3] (97 = the full 1000-frame video; 0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: libraries that print to file descriptor 1 (RCCL's version banner at
    # communicator creation, for one) are sent to stderr for the duration of the run; the line goes out through the saved
    # descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = th.cuda.device_count()          # (does not initialise the GPU)
    if n_dev == 0 or not th.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
    shared_cards = world > n_dev             # rehearsal of N ranks on fewer cards
    if shared_cards:
        # a persistent level chain needs its workgroups co-resident: two ranks' chains on ONE card can starve each other
        # until the bounded waits give up (then the sampler falls back, by design) - a rehearsal runs one launch per stage
        os.environ["LFVDM_LEVEL_CHAIN"] = "0"
    local %= n_dev
    th.cuda.set_device(local)
    dev = th.device("cuda", local)
    backend = None
    # CU budget of the collectives that run beside the backward graph: LFVDM_RCCL_MAX_CHANNELS -> NCCL_MAX_NCHANNELS (each
    # RCCL channel is one workgroup of its reduction kernels), to trade exposed exchange time against backward slowdown
    if os.environ.get("LFVDM_RCCL_MAX_CHANNELS"):
        os.environ["NCCL_MAX_NCHANNELS"] = os.environ["LFVDM_RCCL_MAX_CHANNELS"]
    if world > 1:
        backend = os.environ.get("LFVDM_BENCH_BACKEND") or ("gloo" if shared_cards else "nccl")   # nccl = RCCL
        os.environ["LFVDM_DIST_BACKEND"] = backend
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B, T = 2, 20
    model, diffusion = make_model_and_diffusion(64, dev)
    inputs = synthetic_inputs(B, T, rank, dev)
    shape = (B, T, 4, 16, 16)

    sampler = diffusion._graph_sampler(model, shape, True)
    th.manual_seed(1234 + rank)
    sampler.begin(th.randn(*shape, device=dev), inputs)      # first chain of these weights: capture, tuning, FiLM tables
    cold_setup_ms = float(getattr(sampler, "table_build_ms", 0.0))
    sampler.begin(th.randn(*shape, device=dev), inputs)      # every further chain: only the R tables of its frame indices
    i = diffusion.num_timesteps - 1
    for _ in range(args.warmup):
        sampler.step(i)
        i = max(i - 1, 0)
    k = int(getattr(sampler, "K", 1))
    sampler.run(i, k)                                        # captures the K-steps-per-launch graph outside the timed region
    i = max(i - k, 0)
    times = timed_sampling(sampler, args.steps, world, dev, args.min_seconds, diffusion.num_timesteps)
    regions = len(times)
    # Work that the sampler does once per CHAIN instead of once per step (the R tables: everything that depends on the
    # timestep and this chain's frame indices, GraphSampler.begin) is charged to every timed step at 1/chain-length of
    # its cost.  What is done once per set of WEIGHTS (graph capture, launch tuning, the FiLM tables) is reported as
    # first_chain_setup_ms and, like capture and tuning, not charged.  The same chain through the PUBLIC API
    # (diffusion.p_sample_loop) is timed below and must reproduce `value` (public_api).
    # per-chain work outside the steps (GraphSampler.begin: index tables, the first block of the rolling R window) is
    # charged to every timed step at 1/chain-length; the window's refills run in-stream inside the timed regions
    sampler.begin(th.randn(*shape, device=dev), inputs)
    table_ms = float(sampler.table_build_ms)
    chain_table_ms = float(sampler.chain_table_ms())       # (for the record: every R block of a chain once)
    per_step_s = table_ms * 1e-3 / diffusion.num_timesteps
    el_steps = sum(times)
    el = el_steps + per_step_s * args.steps * regions
    finite = bool(th.isfinite(sampler.plan.x_in).all().item())
    tables_info = (sampler.plan.time_table_bytes, sampler.plan.time_table_fallback)   # fallback: why the per-step plan runs
    # ---- the metric's own definition: one whole chain through the public p_sample_loop (reference
    # gaussian_diffusion.py:403-522: noise draw, per-chain tables, 1000 x p_sample, final clone), wall clock
    th.manual_seed(4321 + rank)
    diffusion.p_sample_loop(model, shape, model_kwargs=inputs, return_decoded=False)        # (warm: same sampler object)
    th.cuda.synchronize()
    walls = []
    for _ in range(2):
        t0 = time.perf_counter()
        pub, _ = diffusion.p_sample_loop(model, shape, model_kwargs=inputs, return_decoded=False)
        th.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    public_wall = min(walls)
    public_api = {"p_sample_loop_wall_ms": round(1000.0 * public_wall, 2), "chain_steps": diffusion.num_timesteps,
                  "steps_per_s_public_api": round(diffusion.num_timesteps / public_wall, 2),
                  "finite": bool(th.isfinite(pub).all().item()),
                  "note": "diffusion.p_sample_loop(model, shape, model_kwargs=..., return_decoded=False), best of 2 warm chains"}
    # persistent level chains of the TIMED plan, and that none of their waits timed out (a timed-out chain leaves garbage in
    # the samples and the sampler falls back to one launch per stage: neither may end up in the line)
    pl_ = sampler.plan
    chains_rec = {"enabled": bool(pl_.chains), "chains": [{"stages": c["n"], "grid": c["grid"], "lds_bytes": c["lds"],
                                                          "work_items": c["items"]} for c in pl_.chains],
                  "launches_replaced": sum(c["n"] for c in pl_.chains) - len(pl_.chains),
                  "timed_out": bool(pl_.chains_aborted()) or bool(getattr(sampler, "chain_timeouts", 0)),
                  "fell_back": bool(getattr(pl_, "chains_off", False)) and os.environ.get("LFVDM_LEVEL_CHAIN", "1") != "0",
                  "off_because_ranks_share_a_gpu": bool(shared_cards)}
    if chains_rec["timed_out"] or chains_rec["fell_back"]:
        raise SystemExit("bench.py: a persistent level chain timed out (LFVDM_CHAIN_TIMEOUT_S) during the timed run")
    train = None
    if args.train_steps > 0:
        del sampler
        diffusion._samplers.clear()
        th.cuda.empty_cache()
        train = bench_train(rank, world, dev, args.train_steps, 4)   # 2 eager + capture + 1 replay before timing
        sampler = diffusion._graph_sampler(model, shape, True)   # for the kernel breakdown below
        sampler.begin(th.randn(*shape, device=dev), inputs)

    total_steps = args.steps * regions
    out = {
        "metric": "denoising steps/sec on 20-frame 4x16x16 latents (sampling; training reported alongside)",
        "value": round(world * total_steps / el, 2),
        "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * el / total_steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timed_regions": regions, "timed_seconds": round(el_steps, 4),
        "p_sample_loop_wall_ms": public_api["p_sample_loop_wall_ms"],
        "steps_per_s_public_api": public_api["steps_per_s_public_api"],
        "public_api": public_api,
        "per_chain_setup": {"begin_table_build_ms": round(table_ms, 3), "table_build_ms": round(table_ms, 3), "first_chain_table_build_ms": round(cold_setup_ms, 3),
                            "chain_steps": diffusion.num_timesteps,
                            "charged_ms_per_step": round(1000.0 * per_step_s, 5),
                            "timestep_table_bytes": int(tables_info[0]), "fallback": tables_info[1],
                            "r_table_ring_steps": int(getattr(sampler.plan, "time_ring", 0)),
                            "r_table_build_ms_per_chain_in_stream": round(chain_table_ms, 3),
                            "note": "value and ms_per_step include the amortised share of begin(); the rolling window's refills "
                                    "(r_table_build_ms_per_chain_in_stream) run inside the timed regions; timed_seconds is the raw time"},
        "region_ms_per_step_min_max": [round(1000.0 * min(times) / args.steps, 4), round(1000.0 * max(times) / args.steps, 4)],
        "config": {"workload": "sample: p_sample loop, latent U-Net num_channels=64 num_res_blocks=1 max_frames=20 "
                               "batch=2 1000-step DDPM on synthetic 4x16x16 latents (BASELINE.json configs[1])",
                   "batch": B, "frames": T, "latent": "4x16x16", "parallelism": f"replicas x{world} (no collective)",
                   "frames_steps_per_s": round(world * total_steps * B * T / el, 1), "finite": finite,
                   "steps_per_graph_launch": int(getattr(sampler, "K", 1)),
                   "gpus_requested": args.gpus},
    }
    out["level_chains"] = chains_rec
    # ---- the collective, self-checked: what the launcher asked for is what the process group is
    rehearsal = world > 1 and backend != "nccl"
    out["collective_world_size"] = dist.get_world_size() if world > 1 else 1
    out["collective_backend"] = ("none (one rank)" if world == 1 else "rccl (torch 'nccl')" if backend == "nccl" else
                                 f"{backend} - REHEARSAL: {world} ranks share {n_dev} GPU(s), not a scaling measurement")
    checks = {"world_size_matches_launcher": out["collective_world_size"] == world,
              "backend_is_rccl_or_labelled_rehearsal": world == 1 or backend == "nccl" or (rehearsal and shared_cards) or
              bool(os.environ.get("LFVDM_BENCH_BACKEND")),
              "ranks_have_their_own_gpu": not shared_cards or rehearsal}
    out["self_check"] = checks
    if not all(checks.values()):
        raise SystemExit(f"bench.py self-check failed: {checks} (WORLD_SIZE={world}, backend={backend}, devices={n_dev})")
    public_api["agrees_with_value_within_3pct"] = abs(public_api["steps_per_s_public_api"] * world / out["value"] - 1.0) <= 0.03
    out["code_stamp"] = running_code_stamp()
    if train is not None and world == 1 and rank == 0 and args.machinery_steps > 0:
        # one GPU can report what the exchange machinery costs with today's code (before any byte crosses xGMI)
        probe = exchange_machinery_probe(dev, args.machinery_steps)
        train["exchange"]["machinery_ms_per_step"] = round(probe["ms_per_step"] - train["ms_per_step"], 3)
        one = probe.get("one_bucket_behind_the_graph_ms_per_step")
        if one is not None:
            # of which the collective itself (one 122 MB "all-reduce" over one rank = a device copy by RCCL) ...
            train["exchange"]["rccl_world1_copy_ms_per_step"] = round(one - train["ms_per_step"], 3)
            # ... and the exchange's own machinery on top of it (5 buckets, 4 of them behind counters inside the backward)
            train["exchange"]["machinery_minus_rccl_copy_ms_per_step"] = round(probe["ms_per_step"] - one, 3)
        train["exchange"]["machinery"] = {"forced_exchange_ms_per_step": probe["ms_per_step"],
                                          "one_bucket_behind_the_graph_ms_per_step": one,
                                          "plain_ms_per_step": train["ms_per_step"], **{k: probe[k] for k in (
                                              "buckets", "buckets_started_inside_the_backward", "buckets_started_after_the_backward",
                                              "overlap_with_backward", "exposed_ms_per_step", "backend", "steps")}}
        train["exchange"]["overlap_probe"] = probe["overlap_probe"]
        train["exchange"]["rccl_max_channels"] = os.environ.get("NCCL_MAX_NCHANNELS")
    if train is not None:
        out["train"] = train
        # the quantity that shards WITH an exchange, at top level and in the same schema at every N
        out["train_videos_per_s"] = train["videos_per_s"]
        out["train_optimizer_steps_per_s"] = train["optimizer_steps_per_s"]
        out["allreduce_bytes_per_step"] = train["allreduce_bytes_per_step"]
        out["exposed_allreduce_ms_per_step"] = train["exchange"]["exposed_ms_per_step"]
        out["exchange"] = {k: train["exchange"].get(k) for k in ("bucket_bytes", "exposed_ms_per_step",
                                                                    "buckets_started_inside_the_backward", "overlap_probe",
                                                                    "machinery_ms_per_step", "rccl_world1_copy_ms_per_step",
                                                                    "machinery_minus_rccl_copy_ms_per_step")}
    if rank == 0:
        # the single-GPU legs (configs[3], configs[4]) and the CPU baselines belong to the N = 1 line only
        if args.long_video_windows > 0 and world == 1:
            out["long_video"] = bench_long_video(dev, args.long_video_windows)
            # the reference's default batch (scripts/video_sample.py:171): eight videos per window, the same bounded leg
            out["long_video"]["batch8"] = bench_long_video(dev, args.long_video_windows, batch=8)
        if args.pixel_steps > 0 and world == 1:
            del sampler
            diffusion._samplers.clear()
            th.cuda.empty_cache()
            out["pixel"] = bench_pixel(dev, args.pixel_steps)
            th.cuda.empty_cache()
            if args.pixel_train_steps > 0:
                # the reference's published training recipe (README.md:54-57) and its rb2 / batch-1 neighbours
                legs = [bench_pixel_train(dev, args.pixel_train_steps, batch=b, num_res_blocks=rb) for b, rb in ((1, 1), (2, 1), (1, 2))]
                out["pixel"]["train"] = legs[0]
                out["pixel"]["train"]["other_configs"] = [{k: lg[k] for k in ("workload", "ms_per_step", "optimizer_steps_per_s",
                                                                               "frames_per_s", "params", "hbm_allocated_peak_gib")}
                                                          | {"tflops": lg["roofline"]["achieved"], "frac": lg["roofline"]["frac"]}
                                                          for lg in legs[1:]]
                th.cuda.empty_cache()
            sampler = diffusion._graph_sampler(model, shape, True)
            sampler.begin(th.randn(*shape, device=dev), inputs)
        if not args.no_breakdown:
            groups = kernel_breakdown(sampler.plan)
            tot_ms = sum(g["ms"] for g in groups.values())
            convs = {k: g for k, g in groups.items() if k.startswith("conv_igemm") or k == "level_chain_kernel"}
            # dominant kernel = the implicit-GEMM template (conv_igemm_kernel): its tile-shape instances together are
            # ~60 % of the step; which instance leads depends on the tuner's picks, so the family is the headline and
            # every instance is listed with what is needed to recompute its fraction
            dom = {"flops": sum(g["flops"] for g in convs.values()), "ms": sum(g["ms"] for g in convs.values()),
                   "launches": sum(g["launches"] for g in convs.values()), "bytes": sum(g["bytes"] for g in convs.values())}
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            # every GEMM FLOP of the step, also the projections that run inside other launches (lfvdm_gn_temporal_qkv)
            gemm_flops = sum(g["flops"] for g in groups.values())
            tr = [(pmc_traffic(k), g["launches"]) for k, g in convs.items()]
            traffic = (round(sum(t * n for t, n in tr if t is not None) / max(1, sum(n for t, n in tr if t is not None)))
                       if any(t is not None for t, _ in tr) else None)
            instances = {}
            for k, g in sorted(convs.items(), key=lambda kv: -kv[1]["ms"]):
                tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
                instances[k] = {"launches": g["launches"], "us": round(1000.0 * g["ms"], 2),
                                "avg_launch_us": round(1000.0 * g["ms"] / g["launches"], 2),
                                "gflop": round(g["flops"] / 1e9, 4), "tflops": round(tf, 2),
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                "algorithmic_bytes": round(g["bytes"] / g["launches"]), "pmc_bytes": pmc_traffic(k)}
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                               "traffic_source": pmc_traffic_source(),
                               "algorithmic_bytes": round(dom["bytes"] / dom["launches"]),
                               "kernel": "the implicit-GEMM body: conv_igemm_kernel (all tile-shape instances) + level_chain_kernel "
                                         "(the same body as stages of the persistent level chains; their small GroupNorm stages ride along)",
                               "launches_per_step": dom["launches"],
                               "avg_launch_us": round(1000.0 * dom["ms"] / dom["launches"], 2),
                               "instances": instances,
                               "note": "fp32 MFMA (shares the vector ALUs with VALU on gfx950: 157.3 TFLOP/s is the peak of both together); "
                                       "achieved = algorithmic conv/GEMM FLOPs of these launches / their HIP-event time; per instance: "
                                       "us = time per step, gflop = FLOPs per step, algorithmic_bytes / pmc_bytes = per launch "
                                       "(operands + output once / PMC L2 fills + write-backs, separate --pmc passes); "
                                       "traffic = launch-weighted mean of pmc_bytes"}
            two_launch = sum(1 for fn, _ in sampler.plan.steps if getattr(fn, "__name__", "") == "lfvdm_gn_apply_ws")
            out["breakdown"] = {"eager_sum_ms": round(tot_ms, 4),
                                "launches": len(sampler.plan.steps) + two_launch + int(getattr(sampler, "extra_launches", 3)),
                                "all_conv_gemm_tflops": round(ach, 2),
                                "step_flops_g": round(gemm_flops / 1e9, 2),
                                "whole_step_frac_of_mfma_peak": round(gemm_flops * out["value"] / world / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                "kernels": {k: {"n": g["launches"], "us": round(1000 * g["ms"], 1)} for k, g in
                                            sorted(groups.items(), key=lambda kv: -kv[1]["ms"])}}
        if world == 1:
            hp = dict(train.get("hbm_phases", {})) if train is not None else {}
            if not args.no_breakdown:
                for key, name in (("gn_apply", "lfvdm_gn_apply"), ("gn_temporal", "lfvdm_gn_temporal")):
                    g = groups.get(name)
                    if g:
                        rec = _hbm_record(g["bytes"] / g["launches"], 1000.0 * g["ms"] / g["launches"],
                                          f"{name}: {g['launches']} launches per cfg-B denoising step, mean bytes (x read, act written) and "
                                          "HIP-event time per launch; 0.16-2.6 MB tensors, L2-resident: launch-latency bound")
                        hp[key] = rec
            hp.update(hbm_phases_latent(diffusion, dev))
            if args.pixel_train_steps > 0 and args.pixel_steps > 0:
                hp.update(hbm_phases_pixel(dev))
                th.cuda.empty_cache()
            out["hbm_phases"] = hp
        if not args.no_cpu and world == 1:
            cores = _host_cores()
            out["cpu_baseline"] = cpu_baseline_sample(model, inputs, B, T, cores, 8.0)
            out["cpu_baseline"]["one_thread"] = cpu_baseline_sample(model, inputs, B, T, 1, 5.0)
            out["cpu_baseline"]["train"] = cpu_baseline_train(cores, 8.0)
            # "sample and train at cfgs A/B/C" (>= 10 timed steps each)
            ma, _ = make_model_and_diffusion(32, th.device("cpu"), steps=32, image_size=32)
            ia = {k: v.cpu() for k, v in synthetic_inputs(1, 5, 0, th.device("cpu")).items()}
            ia["x0"] = th.randn(1, 5, 4, 32, 32, generator=th.Generator().manual_seed(9))
            out["cpu_baseline"]["cfgA_sample"] = cpu_baseline_sample(ma, ia, 1, 5, cores, 4.0, H=32, chain=32,
                                                                     what="configs[0] (32x32, ch32, batch 1, 5 frames, 32-step chain)")
            out["cpu_baseline"]["cfgA_train"] = cpu_baseline_train(cores, 4.0, num_channels=32, B=1, T=5, H=32, chain=32,
                                                                   what="configs[0] (32x32, ch32, batch 1, 5 frames)")
            mc, _ = make_model_and_diffusion(128, th.device("cpu"))
            out["cpu_baseline"]["cfgC_sample"] = cpu_baseline_sample(mc, inputs, B, T, cores, 6.0,
                                                                     what="the configs[2] network (ch128, batch 2, 20 frames)")
            del ma, mc
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():            # (the training leg initialises a world-1 group too)
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
