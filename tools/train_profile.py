"""Developer aid: per-step wall clock of the training leg (cfg C), split into micro-step (graph replay or eager)
and optimizer, for rocprofv3 / wall-clock inspection.  usage: python tools/train_profile.py [steps]"""
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model, diffusion = bench.make_model_and_diffusion(128, dev)
model.train()
loop = TrainLoop(model=model, diffusion=diffusion, data=bench.synthetic_video_stream(2, 40, 4321), batch_size=2,
                 microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                 resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                 schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                 pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
th.manual_seed(99); np.random.seed(99)
orig_fb, orig_opt = loop.forward_backward, loop.optimize_normal
acc = {"fb": 0.0, "opt": 0.0}
def fb(*a, **k):
    th.cuda.synchronize(); t = time.perf_counter(); r = orig_fb(*a, **k); th.cuda.synchronize()
    acc["fb"] = time.perf_counter() - t; return r
def opt(*a, **k):
    th.cuda.synchronize(); t = time.perf_counter(); r = orig_opt(*a, **k); th.cuda.synchronize()
    acc["opt"] = time.perf_counter() - t; return r
loop.forward_backward, loop.optimize_normal = fb, opt
def timed(name):
    orig = getattr(loop, name)
    def f(*a, **k):
        th.cuda.synchronize(); t = time.perf_counter(); r = orig(*a, **k); th.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(loop, name, f)
for nm in ("sample_all_masks", "_graphed_micro_step", "_flush_loss_log"):
    timed(nm)
_replay = th.cuda.CUDAGraph.replay
def replay(self):
    t = time.perf_counter(); _replay(self); acc["replay_host"] = time.perf_counter() - t
th.cuda.CUDAGraph.replay = replay
for i in range(steps):
    for nm in ("sample_all_masks", "_graphed_micro_step", "_flush_loss_log"):
        acc[nm] = 0.0
    th.cuda.synchronize(); t0 = time.perf_counter()
    loop.run_step(); loop.step += 1
    th.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {i}: {dt*1e3:8.2f} ms  fwd+bwd {acc['fb']*1e3:8.2f}  optimizer {acc['opt']*1e3:6.2f}  masks {acc['sample_all_masks']*1e3:6.2f}  micro {acc['_graphed_micro_step']*1e3:6.2f}  log {acc['_flush_loss_log']*1e3:6.2f}  replay(host) {acc.get('replay_host', 0)*1e3:6.2f}", flush=True)
g = loop._graph_state.get("graph") if "--replays" in sys.argv else None     # (extra replays would skew per-step kernel statistics)
if g is not None:
    ts = []
    for _ in range(40):
        th.cuda.synchronize(); t = time.perf_counter(); _replay(g); th.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("bare replays (same inputs):", " ".join(f"{t:.1f}" for t in ts), flush=True)
if g is not None:      # does a replay block the host while the previous replay of the same graph is still running?
    th.cuda.synchronize()
    ts = []
    for _ in range(6):
        t = time.perf_counter(); _replay(g); ts.append((time.perf_counter() - t) * 1e3)
    th.cuda.synchronize()
    print("host time of back-to-back replay calls (no sync):", " ".join(f"{t:.2f}" for t in ts), flush=True)
