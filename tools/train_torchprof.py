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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        loop.run_step(); loop.step += 1
    th.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=70))
