"""Wall time of whole 1000-step chains through the public API (p_sample_loop), tables on/off, cold and warm."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"), ROOT):
    sys.path.insert(0, p)
os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))
import torch as th
import bench
dev = th.device("cuda")
model, diffusion = bench.make_model_and_diffusion(64, dev)
inputs = bench.synthetic_inputs(2, 20, 0, dev)
shape = (2, 20, 4, 16, 16)
for i in range(4):
    th.manual_seed(i)
    th.cuda.synchronize(); t0 = time.perf_counter()
    out, _ = diffusion.p_sample_loop(model, shape, model_kwargs=inputs, return_decoded=False)
    th.cuda.synchronize(); dt = time.perf_counter() - t0
    s = next(iter(diffusion._samplers.values()))
    print(f"tables={os.environ.get('LFVDM_TIME_TABLES','1')} chain {i}: {dt*1e3:8.1f} ms  ({1000/dt:6.1f} steps/s incl. everything)  table_build_ms={s.table_build_ms:.1f}", flush=True)
