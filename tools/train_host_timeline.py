"""Developer aid: host-side time (no synchronisation added) of the phases of TrainLoop.run_step in steady state -
finds the call that blocks the host on the GPU.  usage: python tools/train_host_timeline.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
import argparse as ap
import numpy as np
import torch as th
import bench
from improved_diffusion.train_util import TrainLoop
dev = th.device("cuda")
th.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
model, diffusion = bench.make_model_and_diffusion(128, dev)
model.train()
acc = {}
def data():
    it = bench.synthetic_video_stream(2, 40, 4321)
    while True:
        t = time.perf_counter(); b = next(it); acc.setdefault("next(data)", []).append(time.perf_counter() - t)
        yield b
loop = TrainLoop(model=model, diffusion=diffusion, data=data(), batch_size=2,
                 microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                 resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                 schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                 pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
def timed(obj, name, label=None):
    orig = getattr(obj, name)
    def f(*a, **k):
        t = time.perf_counter(); r = orig(*a, **k); acc.setdefault(label or name, []).append(time.perf_counter() - t); return r
    setattr(obj, name, f)
for nm in ("sample_all_masks", "_upload_async", "_graphed_micro_step", "_flush_loss_log", "_stash_loss_log", "optimize_normal",
           "log_step", "forward_backward", "encode"):
    timed(loop, nm)
timed(loop.arena, "zero_grad", "arena.zero_grad")
timed(loop.schedule_sampler, "sample", "schedule_sampler.sample")
t_steps = []
for i in range(steps):
    t = time.perf_counter(); loop.run_step(); loop.step += 1; t_steps.append(time.perf_counter() - t)
th.cuda.synchronize()
k = steps // 2
print(f"host time per run_step (last {steps - k}): {np.mean(t_steps[k:]) * 1e3:.2f} ms")
for name, v in acc.items():
    per = len(v) / steps
    print(f"  {name:26s} {np.mean(v[int(k * per):]) * 1e3:8.3f} ms x {per:.0f}")
