"""Developer aid: sweep the (frame groups, waves per workgroup) decomposition of the second-generation temporal attention
kernel at the network's shapes; dbg 1 = staging only, 2 = compute only (garbage operands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat
from attn_bench import timeit
L = nat.lib()
dev = th.device("cuda")
shapes = [(2, 20, 256, 64, 4), (2, 20, 64, 128, 4), (2, 20, 4, 128, 4), (2, 20, 256, 128, 4), (1, 14, 256, 64, 4), (2, 20, 64, 256, 4), (2, 20, 16, 256, 4)]
cfgs = [(tg, nw) for tg in (2, 3, 4, 5) for nw in (1, 2, 4)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes, cfgs = shapes[:3], [(4, 4), (4, 2), (5, 4)]
if len(sys.argv) > 1 and sys.argv[1] == "train":        # the cfg-C training shapes with 64-channel heads
    shapes = [(2, 20, 64, 256, 4), (2, 20, 16, 256, 4), (2, 20, 4, 256, 4), (2, 20, 256, 128, 4)]
    cfgs = [(tg, nw) for tg in (2, 4, 5, 7, 10) for nw in (1, 2, 4)]
for (B, T, P, C, heads) in shapes:
    M = B * T * P
    qkv = th.randn(M, 3 * C, device=dev)
    R = [th.randn(B, T, T, C, device=dev) * 0.1 for _ in range(3)]
    mask = (th.rand(B, T, device=dev) < 0.5).float()
    o = th.empty(M, C, device=dev)
    fn = lambda: nat.attn_temporal(qkv, R[0], R[1], R[2], mask, o, None, B, T, P, C, heads)
    L.lfvdm_attn_temporal2_debug(0, 0, 0)
    print(f"B={B} T={T} P={P} C={C}: auto {timeit(fn):6.1f} us", flush=True)
    for tg, nw in cfgs:
        if True:
            row = []
            for dbg in (0, 1, 2, 3):
                L.lfvdm_attn_temporal2_debug(tg, nw, dbg)
                try:
                    row.append(timeit(fn))
                except RuntimeError:
                    row.append(float("nan"))
            print(f"   TG={tg} NW={nw}: full {row[0]:6.1f}  staging-only {row[1]:6.1f}  compute-only {row[2]:6.1f}  compute, all lanes row 0 {row[3]:6.1f}", flush=True)
L.lfvdm_attn_temporal2_debug(0, 0, 0)
