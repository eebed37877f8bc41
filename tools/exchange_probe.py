#!/usr/bin/env python3
"""Developer aid: the cfg-C training step with the bucketed exchange forced on at world size 1 (LFVDM_FORCE_EXCHANGE=1, RCCL),
host-side timeline per step.  usage: python3 tools/exchange_probe.py [steps]"""
import argparse as ap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))
os.environ.setdefault("LFVDM_FORCE_EXCHANGE", "1")
import numpy as np
import torch as th
import bench
from improved_diffusion.train_util import TrainLoop

dev = th.device("cuda", 0)
th.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model, diffusion = bench.make_model_and_diffusion(128, dev)
model.train()
loop = TrainLoop(model=model, diffusion=diffusion, data=bench.synthetic_video_stream(2, 40, 4321), batch_size=2,
                 microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                 resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                 schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                 pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
th.manual_seed(99); np.random.seed(99)
acc = {}


def timed(obj, name, key=None):
    orig = getattr(obj, name)
    key = key or name

    def f(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, f)


timed(loop, "forward_backward")
timed(loop, "optimize_normal")
timed(loop.exchange, "launch", "x.launch")
timed(loop.exchange, "wait", "x.wait")
timed(loop.exchange, "poll_timeout", "x.poll")
for i in range(steps + 5):
    acc.clear()
    if i == 5:
        th.cuda.synchronize(); t_all = time.perf_counter()
    t0 = time.perf_counter()
    loop.run_step(); loop.step += 1
    dt = time.perf_counter() - t0
    print(f"step {i}: host {dt * 1e3:7.2f} ms  " + "  ".join(f"{k} {v * 1e3:6.2f}" for k, v in acc.items()), flush=True)
th.cuda.synchronize()
print(f"wall per step over the last {steps}: {(time.perf_counter() - t_all) / steps * 1e3:.2f} ms; overlap {loop.exchange.overlap}; "
      f"stats {loop.exchange.stats}", flush=True)
