"""Training loop with the reference's interface (train_util.py) on the MI355X-native path.

Differences from the reference, all below the public surface:
  * parameters, gradients, Adam moments and EMA copies each live in ONE flat fp32 arena
    (``ParamArena``); ``model.parameters()`` are views into it, so checkpoints/state-dicts are unchanged;
  * data parallelism = one process per GPU, ONE RCCL all-reduce of the gradient arena per optimizer
    step (after the last micro-batch; the reference gets the same effect from DDP + ``no_sync``,
    train_util.py:116-125,309-313), preceded by one broadcast of the parameter arena from rank 0;
  * ``optimize_normal`` is one fused HIP launch: AdamW + every EMA rate + the gradient norm
    (reference: per-tensor optimizer ops, 2 launches per tensor per EMA rate, 390 ``.item()`` syncs);
  * loss logging does one device->host copy per step instead of one per key and batch element.
"""
import copy
import functools
import glob
import math
import os
from pathlib import Path
from time import time

import numpy as np
import torch as th
import torch.distributed as dist

from . import dist_util
from . import _native as nat
from .logger import logger
from .fp16_util import zero_grad  # noqa: F401  (reference API)
from .nn import update_ema  # noqa: F401  (reference API)
from ._exchange import GradExchange, plan_buckets
from .resample import LossAwareSampler, UniformSampler
from .rng_util import rng_decorator, RNG

try:
    import wandb
except ImportError:  # optional
    wandb = None

INITIAL_LOG_LOSS_SCALE = 20.0


class ParamArena:
    """Flat fp32 storage for a list of parameters: ``p.data`` and ``p.grad`` become views of two
    contiguous buffers, which gives one collective per bucket, one fused optimizer launch and a memset for
    zero_grad.  ``groups`` (lists of parameter indices) fixes the layout order: each group is one contiguous
    slice (``bucket_ranges``) - the unit of the gradient exchange (_exchange.py); default: one group in
    ``named_parameters`` order.  ``self.params`` / ``views()`` keep the caller's order whatever the layout."""

    def __init__(self, params, groups=None):
        self.params = list(params)
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.groups = [list(g) for g in groups] if groups is not None else [list(range(len(self.params)))]
        assert sorted(i for g in self.groups for i in g) == list(range(len(self.params))), "groups must partition the parameters"
        # 16-byte aligned slots so that every parameter view stays float4-addressable for the kernels
        self.offsets, self.bucket_ranges, off = [0] * len(self.params), [], 0
        for g in self.groups:
            lo = off
            for i in g:
                self.offsets[i] = off
                off += (self.sizes[i] + 3) // 4 * 4
            self.bucket_ranges.append((lo, off))
        self.numel = off
        self.p = th.zeros(off, device=dev, dtype=th.float32)
        # four spare floats behind the last bucket: element `numel` is the exchange's skip word (a timed-out bucket wait on
        # ANY rank raises it to 1.0; it rides in the last bucket's SUM all-reduce, _exchange.GradExchange); zero_grad clears it
        self.g_full = th.zeros(off + 4, device=dev, dtype=th.float32)
        self.g = self.g_full[:off]
        with th.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.sizes):
                self.p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.p[o:o + n].view(p.shape)
                p.grad = self.g[o:o + n].view(p.shape)

    def views(self, flat):
        return [flat[o:o + n].view(p.shape) for p, o, n in zip(self.params, self.offsets, self.sizes)]

    def zero_grad(self):
        self.g_full.zero_()
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                p.grad = self.g[o:o + n].view(p.shape)


class TrainLoop:
    def __init__(self, *, model, diffusion, data, batch_size, microbatch, lr, ema_rate, log_interval, save_interval,
                 resume_checkpoint, use_fp16, diffusion_space_kwargs, fp16_scale_growth, schedule_sampler, weight_decay,
                 lr_anneal_steps, sample_interval, pad_with_random_frames, max_frames, enc_dec_chunk_size, args):
        if use_fp16:
            raise NotImplementedError("use_fp16 is off in the reference defaults; the native path is fp32")
        self.args = args
        dist_util.limit_host_threads()
        self.model = model
        self.diffusion = diffusion
        self.data = data
        self.batch_size = batch_size
        self.microbatch = microbatch if microbatch > 0 else batch_size
        self.lr = lr
        self.ema_rate = [ema_rate] if isinstance(ema_rate, float) else [float(x) for x in ema_rate.split(",")]
        self.log_interval = log_interval
        self.save_interval = save_interval
        self.resume_checkpoint = resume_checkpoint
        self.use_fp16 = use_fp16
        self.fp16_scale_growth = fp16_scale_growth
        self.diffusion_space_kwargs = diffusion_space_kwargs
        self.schedule_sampler = schedule_sampler or UniformSampler(diffusion)
        self.weight_decay = weight_decay
        self.lr_anneal_steps = lr_anneal_steps
        self.sample_interval = sample_interval
        self.pad_with_random_frames = pad_with_random_frames
        self.enc_dec_chunk_size = enc_dec_chunk_size
        with RNG(0):
            vis_batch = next(self.data)[0][:2]
            self.vis_batch = self.encode(vis_batch).to(vis_batch.device)
        self.max_frames = max_frames

        self.step = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.global_batch = self.batch_size * self.world
        self.lg_loss_scale = INITIAL_LOG_LOSS_SCALE
        self.sync_cuda = th.cuda.is_available()

        self._load_and_sync_parameters()
        self.model_params = list(self.model.parameters())
        self.master_params = self.model_params
        # gradient arena laid out in the buckets of the data-parallel exchange (backward order, _exchange.py)
        # LFVDM_FORCE_EXCHANGE=1 (testing aid): run the bucketed exchange at world size 1 too - every collective is then a
        # SUM over one rank (the identity), which exercises the whole mechanism on the real backend with a single GPU
        exchanging = self.world > 1 or os.environ.get("LFVDM_FORCE_EXCHANGE") == "1"
        if exchanging and not dist.is_initialized():
            dist_util.setup_dist()
        n_buckets = int(os.environ.get("LFVDM_GRAD_BUCKETS", "5")) if exchanging else 1
        groups, marks = plan_buckets(self.model.named_parameters(), max(1, n_buckets))
        self.arena = ParamArena(self.model_params, groups)
        self.exchange = GradExchange(self.arena, marks, self.world if self.world > 1 else (2 if exchanging else 1))
        if hasattr(self.model, "native_grad_accumulation"):
            self.model.native_grad_accumulation = True      # gradients accumulate in the arena, reduced in optimize_normal
            self.model._grad_exchange = self.exchange if exchanging else None   # bucket markers in the backward pass
        if hasattr(self.model, "_engine"):
            self.model._engine = None          # parameter storage moved: drop cached device pointers
        dev = self.arena.p.device
        self.exp_avg = th.zeros_like(self.arena.p)
        self.exp_avg_sq = th.zeros_like(self.arena.p)
        self.opt_step = 0
        self.betas, self.adam_eps = (0.9, 0.999), 1e-8
        self.opt = _ArenaAdamW(self)          # state_dict()/load_state_dict() in torch.optim.AdamW format
        if getattr(self.args, "resume_id", "") != "":
            self._load_optimizer_state()
            ema_lists = [self._load_ema_parameters(rate) for rate in self.ema_rate]
        else:
            ema_lists = [None for _ in self.ema_rate]
        self.ema_flat = []
        for lst in ema_lists:
            flat = self.arena.p.clone()
            if lst is not None:
                for v, src in zip(self.arena.views(flat), lst):
                    v.copy_(src)
            self.ema_flat.append(flat)
        self.ema_params = [self.arena.views(f) for f in self.ema_flat]
        self.grad_sqsum = th.zeros(1, device=dev)
        self._graph_state = {}

        self.use_ddp = exchanging
        self.ddp_model = self.model            # gradient averaging is done on the arena (see optimize_normal)
        self.exchange.broadcast(self.arena.p, *self.ema_flat)   # same initial replica everywhere (DDP: at construction)
        if self.rank == 0:
            logger.logkv("num_parameters", sum(p.numel() for p in model.parameters()), distributed=False)

    # ------------------------------------------------------------------ checkpoints (resume)
    def _load_and_sync_parameters(self):
        ckpt = find_resume_checkpoint(self.args) or self.resume_checkpoint
        if ckpt:
            self.step = parse_resume_step_from_filename(ckpt)
            print(f"loading model from checkpoint: {ckpt}...")
            self.model.load_state_dict(dist_util.load_state_dict(ckpt, map_location=dist_util.dev())["state_dict"])

    def _load_ema_parameters(self, rate):
        main = find_resume_checkpoint(self.args) or self.resume_checkpoint
        ema_ckpt = find_ema_checkpoint(main, self.step, rate)
        if ema_ckpt:
            print(f"loading EMA from checkpoint: {ema_ckpt}...")
            sd = dist_util.load_state_dict(ema_ckpt, map_location=dist_util.dev())["state_dict"]
            return self._state_dict_to_master_params(sd)
        return None

    def _load_optimizer_state(self):
        main = find_resume_checkpoint(self.args) or self.resume_checkpoint
        if not main:
            return
        path = os.path.join(os.path.dirname(main), f"opt{self.step:06}.pt")
        if os.path.exists(path):
            print(f"loading optimizer state from checkpoint: {path}")
            self.opt.load_state_dict(dist_util.load_state_dict(path, map_location=dist_util.dev()))

    # ------------------------------------------------------------------ training-batch construction (host)
    def sample_some_indices(self, max_indices, T):
        """A random arithmetic-ish progression of <= max_indices frame indices in [0, T)
        (reference train_util.py:180-191; same sequence of random draws).  The reference evaluates
        ``int(pos + i*scale)`` with ``pos`` a float32 0-dim tensor, element by element (~5 us of tensor dispatch each);
        the same float32 arithmetic is done here on a numpy vector."""
        while True:
            s = int(th.randint(low=1, high=max_indices + 1, size=()))
            max_scale = T / (float(s) - 0.999)
            scale = np.exp(np.random.rand() * np.log(max_scale))
            # th.rand(()) * python float: the scalar is rounded to float32, the product is a float32 product
            pos = np.float32(th.rand(()).item()) * np.float32(T - scale * (s - 1))
            indices = (pos + (np.arange(s) * scale).astype(np.float32)).astype(np.int64)     # float32 sums, truncated
            if indices[0] >= 0 and indices[-1] < T and pos + np.float32(0.0) >= 0:
                return indices.tolist()
            print("warning: sampled invalid indices", indices.tolist(), "trying again")

    def _sample_masks_np(self, B, T):
        """Observed / latent frame flags (B, T) of one batch, at most ``max_frames`` flagged per video: the random
        process of reference train_util.py:196-207 with the same sequence of random draws.  Host-side bookkeeping in
        numpy (the reference indexes torch rows element by element, tens of ms per step)."""
        N = self.max_frames
        obs_np = np.zeros((B, T), dtype=np.float32)
        lat_np = np.zeros((B, T), dtype=np.float32)
        for obs_row, latent_row in zip(obs_np, lat_np):
            latent_row[self.sample_some_indices(max_indices=N, T=T)] = 1.
            while True:
                mask = obs_row if th.rand(()) < 0.5 else latent_row
                indices = np.asarray(self.sample_some_indices(max_indices=N, T=T), dtype=np.int64)
                indices = indices[(obs_row[indices] + latent_row[indices]) == 0]
                if len(indices) > N - obs_row.sum() - latent_row.sum():
                    break
                mask[indices] = 1.
        return obs_np, lat_np

    def sample_all_masks(self, batch1, batch2=None, gather=True, set_masks={'obs': (), 'latent': ()}):
        """Random observed/latent frame masks with at most ``max_frames`` frames flagged per video
        (reference train_util.py:193-222)."""
        B, T, *_ = batch1.shape
        obs_np, lat_np = self._sample_masks_np(B, T)
        shape5 = (B, T, 1, 1, 1)
        masks = {'obs': th.from_numpy(obs_np).to(batch1.dtype).view(shape5).to(batch1.device),
                 'latent': th.from_numpy(lat_np).to(batch1.dtype).view(shape5).to(batch1.device)}
        if len(set_masks['obs']) > 0:
            for k in masks:
                n_set = min(len(set_masks[k]), len(masks[k]))
                masks[k][:n_set] = set_masks[k][:n_set]
        any_mask = (masks['obs'] + masks['latent']).to(th.float32).clip(max=1).to(masks['obs'].dtype)
        if not gather:
            return batch1, masks['obs'], masks['latent']
        batch, (obs_mask, latent_mask), frame_indices = self.prepare_training_batch(
            any_mask, batch1, batch2, (masks['obs'], masks['latent']))
        return batch, frame_indices, obs_mask, latent_mask

    def sample_index_table(self, B, T):
        """The training batch as an INDEX TABLE instead of gathered tensors (device-side batch preparation,
        SURVEY 8f.3): int32 (B, max_frames, 4) rows {frame of video1 (< T) or T + frame of the padding video, frame
        index, observed, latent}.  Same random draws, in the same order, as ``sample_all_masks`` followed by
        ``prepare_training_batch`` with ``pad_with_random_frames`` (reference :196-207 then :236 per video), so a seed
        gives the same batch on either path; like the reference, a padding slot inherits the flags of the frame
        position it was drawn at (:239-241)."""
        assert self.pad_with_random_frames, "the index table describes fixed-size batches (pad_with_random_frames)"
        F = self.max_frames
        obs_np, lat_np = self._sample_masks_np(B, T)
        table = np.zeros((B, F, 4), dtype=np.int32)
        for b in range(B):
            flagged = np.flatnonzero((obs_np[b] + lat_np[b]) > 0)
            n = len(flagged)
            pads = th.randint_like(th.zeros(F - n, dtype=th.int64), high=T).numpy()
            frames = np.concatenate([flagged, pads])
            table[b, :, 0] = np.concatenate([flagged, T + pads])
            table[b, :, 1] = frames
            table[b, :, 2] = obs_np[b, frames]
            table[b, :, 3] = lat_np[b, frames]
        return table

    def prepare_training_batch(self, mask, batch1, batch2, tensors):
        """Gather the flagged frames of batch1 (sorted), pad to ``max_frames`` with random frames of
        batch2 and gather ``tensors`` the same way (reference train_util.py:224-241)."""
        B, T, *_ = mask.shape
        mask = mask.view(B, T)
        eff_T = self.max_frames if self.pad_with_random_frames else int(mask.sum(dim=1).max())
        indices = th.zeros_like(mask[:, :eff_T], dtype=th.int64)
        new_batch = th.zeros_like(batch1[:, :eff_T])
        new_tensors = [th.zeros_like(t[:, :eff_T]) for t in tensors]
        for b in range(B):
            n = int(mask[b].sum())
            indices[b, :n] = mask[b].nonzero().flatten()
            indices[b, n:] = th.randint_like(indices[b, n:], high=T) if self.pad_with_random_frames else 0
            new_batch[b, :n] = batch1[b][mask[b] == 1]
            new_batch[b, n:] = (batch1 if batch2 is None else batch2)[b][indices[b, n:]]
            for nt, t in zip(new_tensors, tensors):
                nt[b, :n] = t[b][mask[b] == 1]
                nt[b, n:] = t[b][indices[b, n:]]
        return new_batch, new_tensors, indices

    # ------------------------------------------------------------------ loop
    def run_loop(self):
        last_sample_time = None
        while not self.lr_anneal_steps or self.step < self.lr_anneal_steps:
            self.run_step()
            if self.step % self.log_interval == 0:
                logger.dumpkvs()
            if self.step % self.save_interval == 0:
                self.save()
            if os.environ.get("DIFFUSION_TRAINING_TEST", "") and self.step > 0:
                return
            if self.sample_interval is not None and self.step != 0 and (self.step % self.sample_interval == 0 or self.step == 5):
                if last_sample_time is not None:
                    logger.logkv('timing/time_between_samples', time() - last_sample_time)
                self.log_samples()
                last_sample_time = time()
            self.step += 1
        if (self.step - 1) % self.save_interval != 0:
            self.save()

    def run_step(self):
        t0 = time()
        self.forward_backward()
        self.optimize_normal()
        self.log_step()
        logger.logkv("timing/step_time", time() - t0)

    def _micro_step(self, micro, frame_indices, obs_mask, latent_mask, t, weights):
        """q_sample -> U-Net forward -> masked MSE -> backward for one micro-batch (device tensors in, the
        per-sample loss terms out).  No host synchronisation inside: the body is hipGraph-capturable."""
        losses = self.diffusion.training_losses(
            self.ddp_model, micro, t,
            model_kwargs={'frame_indices': frame_indices, 'obs_mask': obs_mask, 'latent_mask': latent_mask, 'x0': micro},
            latent_mask=(1 - obs_mask) if self.pad_with_random_frames else latent_mask, eval_mask=latent_mask)
        # reference train_util.py:320-328: loss = (losses["loss"] * weights).mean(); loss.backward().  d loss / d loss[b] =
        # weights[b] / B: the backward pass is seeded with it directly (the scalar's own
        # forward and backward were five tiny launches); gradients accumulate in the arena across micro-batches
        losses["loss"].backward(gradient=weights.to(losses["loss"].dtype) / weights.numel())
        weighted, seen = {}, {}
        for k, v in losses.items():          # ("loss" is the very tensor of "mse": weight each distinct term once)
            if id(v) not in seen:
                seen[id(v)] = (v * weights).detach()
            weighted[k] = seen[id(v)]
        return weighted, losses["loss"].detach()

    def _micro_step_from_pool(self, pool, table, t, weights):
        """Device-side batch preparation + micro-step: gather the frames named by the index table and write the mask /
        index tensors (lfvdm_prepare_batch), then the usual micro-step.  One capturable body."""
        B, F = table.shape[0], table.shape[1]
        micro = th.empty((B, F) + tuple(pool.shape[2:]), device=pool.device, dtype=th.float32)
        frame_indices = th.empty(B, F, device=pool.device, dtype=th.int64)
        obs_mask = th.empty(B, F, 1, 1, 1, device=pool.device, dtype=th.float32)
        latent_mask = th.empty_like(obs_mask)
        nat.prepare_batch(pool, table, micro, frame_indices, obs_mask, latent_mask)
        return self._micro_step(micro, frame_indices, obs_mask, latent_mask, t, weights)

    def _graphed_micro_step(self, inputs, body=None, upload=None):
        """Replay the captured micro-step (forward + backward, ~1000 launches) as ONE hipGraph: the training
        step is host-bound otherwise.  Captured once the shapes have been seen twice; static input buffers.
        ``body``: the capturable function of the inputs (default ``_micro_step``).  ``upload``: (pinned bytes, views) of
        inputs that are still on the host - ``inputs`` is then None and the bytes are copied straight into the static
        buffers (one H2D copy, no staging tensor on the device)."""
        body = body or self._micro_step
        sig = [(shape, dt) for _, _, dt, shape in upload[1]] if upload is not None else [(tuple(x.shape), x.dtype) for x in inputs]
        key = (body.__name__,) + tuple(sig)
        st = self._graph_state
        if st.get("key") != key:
            st.clear()
            st.update(key=key, seen=0)
        st["seen"] += 1
        if st["seen"] <= 2 or os.environ.get("LFVDM_TRAIN_GRAPH", "1") == "0":
            if upload is not None:
                inputs = self._to_device(upload)
            self._dev_inputs = inputs
            return body(*inputs)          # eager warm-up (also sets kernel attributes)
        # A replay runs no Python, so the version-gated re-pack of the conv weights inside the autograd blocks never
        # fires there: bring the packed copies up to date eagerly, before the capture (which then records no pack
        # launch, whatever micro-batch of the optimizer step it happens to land on) and before every replay.
        from ._backward import _packs
        _packs.refresh_if_stale()
        if "graph" not in st:
            if upload is not None:
                stage, views = upload
                st["static_bytes"] = th.empty(stage.numel(), device=dist_util.dev(), dtype=th.uint8)
                st["static_in"] = [st["static_bytes"][o:o + nb].view(dt).view(shape) for o, nb, dt, shape in views]
                st["static_bytes"].copy_(stage, non_blocking=True)
            else:
                st["static_in"] = [x.clone() for x in inputs]
            th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            # thread-local capture: the RCCL watchdog thread queries the events of earlier collectives meanwhile
            with th.cuda.graph(g, capture_error_mode="thread_local"):
                st["static_out"] = body(*st["static_in"])
            st["graph"] = g
            # the capture itself does not execute: run the step for real below
        if upload is not None:
            st["static_bytes"].copy_(upload[0], non_blocking=True)
            self._stage_issued()
        else:
            for dst, src in zip(st["static_in"], inputs):
                dst.copy_(src)
        st["graph"].replay()
        self._dev_inputs = st["static_in"]
        return st["static_out"]

    def _device_prep_ok(self, batch1):
        """Device-side batch preparation applies to fixed-size batches of host tensors whose encode step is the identity
        (pixel space or pre-encoded latents); LFVDM_DEVICE_BATCH_PREP=0 restores the host gather."""
        d = self.diffusion
        return (self.pad_with_random_frames and not batch1.is_cuda and dist_util.dev().type == "cuda"
                and batch1.dtype == th.float32 and batch1[0, 0].numel() % 4 == 0
                and (getattr(d, "diffusion_space", None) in (None, "pixel") or bool(getattr(d, "pre_encoded", False)))
                and os.environ.get("LFVDM_DEVICE_BATCH_PREP", "1") != "0")

    # videos up to this many bytes travel whole and are gathered on the device; longer ones (CARLA: 1000 frames) are
    # thinned on the host to the frames the table names, so that only those cross PCIe
    POOL_WHOLE_VIDEO_BYTES = 2 << 20

    def _pool_and_table(self, micro1, micro2, table):
        """(pool, table) for lfvdm_prepare_batch: the whole videos [video1 | padding video] when they are short, else
        the named frames only (one vectorised index_select per video; table rows renumbered 0..F-1)."""
        B, T = micro1.shape[:2]
        src2 = micro1 if micro2 is None else micro2
        if 2 * T * micro1[0, 0].numel() * 4 <= self.POOL_WHOLE_VIDEO_BYTES:
            return th.cat([micro1, src2], dim=1), table
        rows = th.from_numpy(table[:, :, 0].astype(np.int64))
        pool = th.stack([th.where((rows[b] < T).view(-1, *([1] * (micro1.dim() - 2))),
                                  micro1[b].index_select(0, rows[b].clamp(max=T - 1)),
                                  src2[b].index_select(0, (rows[b] - T).clamp(min=0))) for b in range(B)])
        table = table.copy()
        table[:, :, 0] = np.arange(table.shape[1], dtype=np.int32)[None]
        return pool, table

    def forward_backward(self):
        self.arena.zero_grad()
        batch1 = next(self.data)[0]
        batch2 = next(self.data)[0] if self.pad_with_random_frames else None
        for i in range(0, batch1.shape[0], self.microbatch):
            micro1 = batch1[i:i + self.microbatch]
            micro2 = batch2[i:i + self.microbatch] if batch2 is not None else None
            dev = dist_util.dev()
            if self._device_prep_ok(micro1):
                # host: the index table (same random draws as the reference's mask sampling); device: everything else
                table = self.sample_index_table(micro1.shape[0], micro1.shape[1])
                pool, table = self._pool_and_table(micro1, micro2, table)
                t, weights = self.schedule_sampler.sample(micro1.shape[0], th.device("cpu"))
                upload = self._stage(pool, th.from_numpy(table), t, weights)
                weighted, raw = self._graphed_micro_step(None, body=self._micro_step_from_pool, upload=upload)
                t = self._dev_inputs[2]            # the timesteps as the device saw them (loss-quartile logging)
            else:
                micro, frame_indices, obs_mask, latent_mask = self.sample_all_masks(micro1, micro2)
                micro = self.encode(micro)
                if dev.type == "cuda" and not micro.is_cuda:
                    t, weights = self.schedule_sampler.sample(micro.shape[0], th.device("cpu"))
                    inputs = self._to_device(self._stage(micro, frame_indices, obs_mask, latent_mask, t, weights))
                    micro, frame_indices, obs_mask, latent_mask, t, weights = inputs
                else:
                    micro, frame_indices = micro.to(dev), frame_indices.to(dev)
                    obs_mask, latent_mask = obs_mask.to(dev), latent_mask.to(dev)
                    t, weights = self.schedule_sampler.sample(micro.shape[0], dev)
                    inputs = (micro, frame_indices, obs_mask, latent_mask, t, weights)
                if micro.is_cuda and self.pad_with_random_frames:
                    weighted, raw = self._graphed_micro_step(inputs)
                else:
                    weighted, raw = self._micro_step(*inputs)
            self.exchange.micro_step_done()
            if isinstance(self.schedule_sampler, LossAwareSampler):
                self.schedule_sampler.update_with_local_losses(t, raw)
            # The loss terms are logged one micro-step late (or at the next dumpkvs/save): reading them now
            # would stall the host on the GPU and serialise batch preparation with the device work.
            self._flush_loss_log()
            self._stash_loss_log(t, weighted)

    def _stage(self, *host_tensors):
        """All host inputs of a micro-step packed into ONE pinned buffer of a staging ring -> (pinned bytes, views):
        a pageable ``.to(device)`` blocks the host until the GPU has drained the previous step, which would
        serialise batch preparation with the device work."""
        ring = self.__dict__.setdefault("_stage_ring", {"slots": [], "next": 0})
        sizes = [(x.numel() * x.element_size() + 15) // 16 * 16 for x in host_tensors]
        total = sum(sizes)
        if not ring["slots"] or ring["slots"][0][0].numel() < total:
            ring["slots"] = [[th.empty(total, dtype=th.uint8).pin_memory(), None] for _ in range(4)]
            ring["next"] = 0
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % len(ring["slots"])
        if slot[1] is not None:
            self._blocked(slot[1].synchronize)       # copy issued 4 micro-steps ago: long finished
            slot[1] = None
        stage, off, views = slot[0], 0, []
        for x, n in zip(host_tensors, sizes):
            nb = x.numel() * x.element_size()
            stage[off:off + nb].view(x.dtype).copy_(x.reshape(-1))
            views.append((off, nb, x.dtype, tuple(x.shape)))
            off += n
        self._stage_slot = slot
        return stage[:total], views

    def _stage_issued(self):
        """The H2D copy of the most recently staged buffer has been enqueued: mark when the slot may be reused."""
        ev = th.cuda.Event()
        ev.record()
        self._stage_slot[1] = ev

    def _to_device(self, upload):
        """Asynchronous copy of a staged buffer into a fresh device tensor -> tuple of device views."""
        stage, views = upload
        dbuf = stage.to(dist_util.dev(), non_blocking=True)
        self._stage_issued()
        return tuple(dbuf[o:o + nb].view(dt).view(shape) for o, nb, dt, shape in views)

    def _blocked(self, fn):
        """Run a call that may block on the GPU and account the time (``host_wait_s``): the rest of a step is host WORK."""
        t0 = time()
        out = fn()
        self.host_wait_s = getattr(self, "host_wait_s", 0.0) + (time() - t0)
        return out

    def _stash_loss_log(self, t, weighted):
        """Queue the per-sample loss terms for logging without stalling this stream: a side stream copies them
        into pinned memory once they are computed; ``_flush_loss_log`` reads them one micro-step later."""
        keys = list(weighted.keys())
        if not t.is_cuda:
            self._pending_log = (keys, th.stack([weighted[k].detach().float() for k in keys] + [t.float()]), None)
            return
        packed = th.stack([weighted[k].detach().float() for k in keys] + [t.float()])
        side = self.__dict__.setdefault("_log_stream", th.cuda.Stream())
        ready = th.cuda.Event()
        ready.record()
        ring = self.__dict__.setdefault("_log_ring", {"bufs": [], "next": 0})
        if not ring["bufs"] or ring["bufs"][0].shape != packed.shape:
            ring["bufs"] = [th.empty(packed.shape, dtype=th.float32).pin_memory() for _ in range(4)]
        host = ring["bufs"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % 4
        with th.cuda.stream(side):
            side.wait_event(ready)
            host.copy_(packed, non_blocking=True)
            packed.record_stream(side)
            done = th.cuda.Event()
            done.record(side)
        self._pending_log = (keys, host, done)

    def _flush_loss_log(self):
        if self._flush_loss_log not in logger.pre_dump_hooks:
            logger.pre_dump_hooks[:] = [h for h in logger.pre_dump_hooks
                                        if getattr(h, "__func__", None) is not TrainLoop._flush_loss_log]
            logger.pre_dump_hooks.append(self._flush_loss_log)
        pending, self._pending_log = getattr(self, "_pending_log", None), None
        if pending is not None:
            keys, host, done = pending
            if done is not None:
                self._blocked(done.synchronize)
            vals = host.numpy()
            log_loss_dict(self.diffusion, vals[-1], {k: vals[i] for i, k in enumerate(keys)})

    def optimize_normal(self):
        """Bucketed all-reduce (overlapped with the tail of the backward pass) + fused AdamW/EMA/grad-norm
        (reference train_util.py:346-357; the exchange is DDP's there, :116-125)."""
        if self.use_ddp:
            self.exchange.launch()      # RCCL SUM per bucket on the side stream; averaged by grad_scale below
            self.exchange.wait()
        self._anneal_lr()
        self.opt_step += 1
        self.grad_sqsum.zero_()
        a = nat.AdamWArgs()
        a.p, a.g, a.m, a.v = self.arena.p.data_ptr(), self.arena.g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        for i, (flat, rate) in enumerate(zip(self.ema_flat, self.ema_rate)):
            a.ema[i] = flat.data_ptr()
            a.ema_rate[i] = rate
        a.n_ema, a.n = len(self.ema_flat), self.arena.numel
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = self.cur_lr(), self.betas[0], self.betas[1], self.adam_eps, self.weight_decay
        a.bias_corr1 = 1.0 - self.betas[0] ** self.opt_step
        a.bias_corr2_sqrt = math.sqrt(1.0 - self.betas[1] ** self.opt_step)
        a.grad_scale = 1.0 / self.world
        a.grad_sqsum = self.grad_sqsum.data_ptr()
        a.skip_flag = self.exchange.skip_flag_ptr()     # a timed-out bucket wait turns this launch into a no-op ...
        a.skip_flag2 = self.exchange.skip_word_ptr()    # ... on every rank: the word that rode in the last bucket
        import ctypes
        nat.check(nat.lib().lfvdm_adamw_ema(ctypes.byref(a), nat.stream()), "lfvdm_adamw_ema")
        self._invalidate_engine()
        try:
            self.exchange.poll_timeout()         # every step: the words of the PREVIOUS step, read through pinned memory
        except RuntimeError:
            self.opt_step -= getattr(self.exchange, "skipped_steps", 0)      # bias correction counts applied steps only
            raise
        if self.step % self.log_interval == 0:   # the only host sync of the optimizer phase
            logger.logkv_mean("grad_norm", float(np.sqrt(self.grad_sqsum.item())))
            self.exchange.poll_timeout(sync=True)

    def _invalidate_engine(self):
        nat.param_epoch[0] += 1          # cached packed weights (sampler plans, training path) are stale now
        eng = getattr(self.model, "_engine", None)
        if eng is not None:
            eng.invalidate()

    def cur_lr(self):
        if not self.lr_anneal_steps:
            return self.lr
        return self.lr * (1 - self.step / self.lr_anneal_steps)

    def _anneal_lr(self):
        self.opt.param_groups[0]["lr"] = self.cur_lr()

    def log_step(self):
        logger.logkv("step", self.step)
        logger.logkv("samples", (self.step + 1) * self.global_batch)

    # ------------------------------------------------------------------ save
    def save(self):
        if self.rank == 0:
            Path(get_blob_logdir(self.args)).mkdir(parents=True, exist_ok=True)

            def save_checkpoint(rate, params):
                print(f"saving model {rate}...")
                name = f"model{self.step:06d}.pt" if not rate else f"ema_{rate}_{self.step:06d}.pt"
                th.save({"state_dict": self._master_params_to_state_dict(params), "config": self.args.__dict__,
                         "step": self.step}, os.path.join(get_blob_logdir(self.args), name))

            save_checkpoint(0, self.master_params)
            for rate, params in zip(self.ema_rate, self.ema_params):
                save_checkpoint(rate, params)
            th.save(self.opt.state_dict(), os.path.join(get_blob_logdir(self.args), f"opt{self.step:06d}.pt"))
        if dist.is_initialized():
            dist.barrier()

    def _master_params_to_state_dict(self, master_params):
        sd = self.model.state_dict()
        for i, (name, _v) in enumerate(self.model.named_parameters()):
            assert name in sd
            sd[name] = master_params[i].detach().clone()
        return sd

    def _state_dict_to_master_params(self, state_dict):
        return [state_dict[name] for name, _ in self.model.named_parameters()]

    def encode(self, video):
        return self.diffusion.encode(video, chunk_size=self.enc_dec_chunk_size)

    def decode(self, video):
        return self.diffusion.decode(video, chunk_size=self.enc_dec_chunk_size)

    @rng_decorator(seed=0)
    def log_samples(self):
        """Rank-0 sampling with the first EMA parameter set (reference train_util.py:428-475)."""
        if self.rank == 0:
            sample_start = time()
            self.model.eval()
            with th.no_grad():
                backup = self.arena.p.clone()
                self.arena.p.copy_(self.ema_flat[0])
            self._invalidate_engine()
            print("sampling...")
            obs_mask = th.zeros_like(self.vis_batch[:, :, :1, :1, :1])
            latent_mask = obs_mask.clone()
            n_obs = self.max_frames // 3
            obs_mask[0, :n_obs] = 1.
            latent_mask[0, n_obs:self.max_frames] = 1.
            if self.batch_size > 1 and len(self.vis_batch) > 1:
                spacing = len(self.vis_batch[0]) // self.max_frames
                obs_mask[1, :n_obs * spacing:spacing] = 1.
                latent_mask[1, n_obs * spacing:self.max_frames * spacing:spacing] = 1.
            batch, frame_indices, obs_mask, latent_mask = self.sample_all_masks(
                self.vis_batch, None, gather=True, set_masks={'obs': obs_mask, 'latent': latent_mask})
            dev = dist_util.dev()
            samples, attn = self.diffusion.p_sample_loop(
                self.model, batch.shape, clip_denoised=True,
                model_kwargs={'frame_indices': frame_indices.to(dev), 'x0': batch.to(dev), 'obs_mask': obs_mask.to(dev),
                              'latent_mask': latent_mask.to(dev)},
                latent_mask=latent_mask, return_attn_weights=False, return_decoded=False)
            samples = samples.cpu() * latent_mask + batch * obs_mask
            renderable = self.diffusion.diffusion_space in (None, "pixel") or self.diffusion.vae is not None
            if renderable:     # latent space without an attached VAE: nothing to render, only the timing is logged
                samples = self.decode(samples).float()
                _mark_as_observed(samples[:, :n_obs])
                vids = ((samples + 1) * 127.5).clamp(0, 255).to(th.uint8).cpu().numpy()
                if wandb is not None and wandb.run is not None:
                    for i, video in enumerate(vids):
                        logger.logkv(f'video-{i}', wandb.Video(video), distributed=False)
            logger.logkv("timing/sampling_time", time() - sample_start, distributed=False)
            self.model.train()
            with th.no_grad():
                self.arena.p.copy_(backup)
            self._invalidate_engine()
            print("finished sampling")
        if dist.is_initialized():
            dist.barrier()


class _ArenaAdamW:
    """torch.optim.AdamW-compatible view of the arena optimizer state (for checkpoints and lr access)."""

    def __init__(self, loop):
        self.loop = loop
        self.param_groups = [dict(lr=loop.lr, betas=loop.betas, eps=loop.adam_eps, weight_decay=loop.weight_decay,
                                  amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=None, params=list(range(len(loop.model_params))))]

    def state_dict(self):
        L = self.loop
        m, v = L.arena.views(L.exp_avg), L.arena.views(L.exp_avg_sq)
        state = {i: {"step": th.tensor(float(L.opt_step)), "exp_avg": m[i].clone(), "exp_avg_sq": v[i].clone()}
                 for i in range(len(L.model_params))} if L.opt_step > 0 else {}
        return {"state": state, "param_groups": [dict(self.param_groups[0])]}

    def load_state_dict(self, sd):
        L = self.loop
        m, v = L.arena.views(L.exp_avg), L.arena.views(L.exp_avg_sq)
        for i, st in sd.get("state", {}).items():
            m[int(i)].copy_(st["exp_avg"])
            v[int(i)].copy_(st["exp_avg_sq"])
            L.opt_step = int(float(st["step"]))


def _mark_as_observed(images, color=[1., -1., -1.]):
    for i, c in enumerate(color):
        if i >= images.shape[-3]:
            break
        images[..., i, :, 1:2] = c
        images[..., i, 1:2, :] = c
        images[..., i, :, -2:-1] = c
        images[..., i, -2:-1, :] = c


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNN.pt -> NNNNNN (0 if it does not parse)."""
    parts = filename.split("model")
    if len(parts) < 2:
        return 0
    try:
        return int(parts[-1].split(".")[0])
    except ValueError:
        return 0


def get_blob_logdir(args):
    root_dir = "checkpoints"
    assert os.path.exists(root_dir), "Must create directory 'checkpoints'"
    if len(getattr(args, "resume_id", "")) > 0:
        run_id = args.resume_id
    elif wandb is not None and wandb.run is not None:
        run_id = wandb.run.id
    else:
        run_id = os.environ.get("LFVDM_RUN_ID", "local")
    return os.path.join(root_dir, run_id)


def find_resume_checkpoint(args):
    if not getattr(args, "resume_id", ""):
        return None
    ckpts = glob.glob(os.path.join(get_blob_logdir(args), "model*.pt"))
    if not ckpts:
        return None
    by_step = {int(Path(f).stem.replace('model', '')): f for f in ckpts}
    return by_step[max(by_step)]


def find_ema_checkpoint(main_checkpoint, step, rate):
    if main_checkpoint is None:
        return None
    path = os.path.join(os.path.dirname(main_checkpoint), f"ema_{rate}_{step:06d}.pt")
    return path if os.path.exists(path) else None


def log_loss_dict(diffusion, ts, losses):
    """Mean and per-timestep-quartile means of every loss term (reference train_util.py:530-536), with a
    single device->host transfer."""
    keys = list(losses.keys())
    if isinstance(ts, np.ndarray):       # already on the host (TrainLoop's deferred logging)
        stacked, ts_np = np.stack([np.asarray(losses[k], dtype=np.float32) for k in keys]), ts
    else:
        stacked = th.stack([losses[k].detach().float() for k in keys]).cpu().numpy()
        ts_np = ts.cpu().numpy()
    for key, values in zip(keys, stacked):
        logger.logkv_mean(key, float(values.mean()))
        for sub_t, sub_loss in zip(ts_np, values):
            quartile = int(4 * sub_t / diffusion.num_timesteps)
            logger.logkv_mean(f"{key}_q{quartile}", float(sub_loss))
