"""Developer experiment: one captured forward at batch 2 vs two concurrent batch-1 forwards (two streams forked
inside one graph).  The chain of ~160 dependent launches is latency-bound; do two independent chains overlap?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))
import torch as th
import bench
from improved_diffusion._engine import Plan

dev = th.device("cuda")
model, diffusion = bench.make_model_and_diffusion(64, dev)
T = 20
inp = bench.synthetic_inputs(2, T, 0, dev)
eng = model.native_engine()


def make(B, sl):
    pl = Plan(eng, B, T, 16, 16, False)
    pl.refresh_weights()
    pl.set_inputs(th.randn(B, T, 4, 16, 16, device=dev), inp["x0"][sl], th.full((B,), 500.0, device=dev),
                  inp["frame_indices"][sl], inp["obs_mask"][sl], inp["latent_mask"][sl])
    pl.launch(); pl.autotune(); return pl


def timeit(g, n=300):
    for _ in range(20): g.replay()
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    th.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6


full = make(2, slice(0, 2))
halves = [make(1, slice(0, 1)), make(1, slice(1, 2))]
th.cuda.synchronize()
g1 = th.cuda.CUDAGraph()
with th.cuda.graph(g1):
    full.launch()
g2 = th.cuda.CUDAGraph()
with th.cuda.graph(g2):
    cur = th.cuda.current_stream()
    side = [th.cuda.Stream(), th.cuda.Stream()]
    for s, pl in zip(side, halves):
        s.wait_stream(cur)
        with th.cuda.stream(s):
            pl.launch()
    for s in side:
        cur.wait_stream(s)
g3 = th.cuda.CUDAGraph()
with th.cuda.graph(g3):
    halves[0].launch()
g4 = th.cuda.CUDAGraph()
with th.cuda.graph(g4):
    halves[0].launch(); halves[1].launch()
print(f"batch 2, one chain          : {timeit(g1):8.1f} us")
print(f"batch 1, one chain          : {timeit(g3):8.1f} us")
print(f"2 x batch 1, sequential     : {timeit(g4):8.1f} us")
print(f"2 x batch 1, two streams    : {timeit(g2):8.1f} us")
