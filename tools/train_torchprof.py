"""Developer aid: attribute the remaining torch-library kernels of an eager training step to ATen ops (with shapes)."""
import os, sys
os.environ["LFVDM_TRAIN_GRAPH"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
import argparse as ap
import numpy as np
import torch as th
import bench
from improved_diffusion.train_util import TrainLoop
dev = th.device("cuda"); th.cuda.set_device(0)
model, diffusion = bench.make_model_and_diffusion(128, dev); model.train()
loop = TrainLoop(model=model, diffusion=diffusion, data=bench.synthetic_video_stream(2, 40, 4321), batch_size=2,
                 microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                 resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                 schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                 pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
th.manual_seed(99); np.random.seed(99)
for _ in range(3):
    loop.run_step(); loop.step += 1
th.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(2):
        loop.run_step(); loop.step += 1
    th.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=70))
# the library (ATen) launches only, with the Python frames that issued them
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
        if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:45]:
    frames = [f for f in e.stack if "improved_diffusion" in f or "bench.py" in f][:3]
    print(f"{e.key:28s} n={e.count:3d} cuda={e.self_device_time_total:8.1f}us shapes={str(e.input_shapes)[:60]:60s} | " + " <- ".join(
        f.split("improved_diffusion/")[-1][:60] for f in frames))
