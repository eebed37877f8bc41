"""Developer aid: which ATen (library) ops does one eager training micro-step at cfg C still issue, and from where?
Runs the micro-step eagerly (LFVDM_TRAIN_GRAPH=0) under a TorchDispatchMode and prints every dispatched op that
launches a kernel, grouped by (op, innermost improved_diffusion frame).  usage: python tools/train_aten_ops.py"""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ["LFVDM_TRAIN_GRAPH"] = "0"
import argparse as ap
import numpy as np
import torch as th
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from improved_diffusion.train_util import TrainLoop

dev = th.device("cuda")
model, diffusion = bench.make_model_and_diffusion(128, dev)
model.train()
loop = TrainLoop(model=model, diffusion=diffusion, data=bench.synthetic_video_stream(2, 40, 4321), batch_size=2,
                 microbatch=-1, lr=1e-4, ema_rate="0.9999", log_interval=10 ** 9, save_interval=10 ** 9,
                 resume_checkpoint="", use_fp16=False, diffusion_space_kwargs={}, fp16_scale_growth=1e-3,
                 schedule_sampler=None, weight_decay=0.0, lr_anneal_steps=0, sample_interval=None,
                 pad_with_random_frames=True, max_frames=20, enc_dec_chunk_size=20, args=ap.Namespace(resume_id=""))
th.manual_seed(99); np.random.seed(99)
for _ in range(3):
    loop.run_step(); loop.step += 1
th.cuda.synchronize()

VIEWS = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
         "as_strided", "t.default", "unbind", "split", "_unsafe_view", "empty", "is_", "size", "stride", "numel", "sym_",
         "_local_scalar", "lift_fresh", "_to_copy_meta", "storage_offset", "is_contiguous", "unflatten", "flatten", "narrow",
         "new_empty", "result_type", "can_cast", "prim.", "dim.")
count = collections.Counter()
shapes = {}


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEWS):
            where = "(autograd engine / no repo frame)"
            for fr in reversed(traceback.extract_stack()):
                if "improved_diffusion" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            key = (name, where)
            count[key] += 1
            o = out[0] if isinstance(out, (tuple, list)) and out else out
            if isinstance(o, th.Tensor):
                shapes.setdefault(key, tuple(o.shape))
        return out


with Log():
    loop.run_step()       # forward + backward AND the optimizer phase
th.cuda.synchronize()
tot = 0
for (name, where), n in sorted(count.items(), key=lambda kv: (-kv[1], kv[0])):
    tot += n
    print(f"{n:4d}  {name:40s} {str(shapes.get((name, where), '')):28s} {where}")
print("total dispatched non-view ops:", tot)
