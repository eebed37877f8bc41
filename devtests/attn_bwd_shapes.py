"""Developer aid: rocprofv3-friendly loop over the temporal-attention backward at the cfg-C shapes (kernel times per shape
are read from the kernel trace: one shape per process, `python devtests/attn_bwd_shapes.py P C`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat
B, T, heads = 2, 20, 4
for P, C in ((256, 128), (64, 256), (16, 256), (4, 256)):
    qkv = th.randn(B * T * P, 3 * C, device="cuda")
    d_o = th.randn(B * T * P, C, device="cuda")
    R = [th.randn(B, T, T, C, device="cuda") * 0.1 for _ in range(3)]
    mask = th.ones(B, T, device="cuda")
    ws_p = th.empty(B * P * heads, T, T, device="cuda"); ws_ds = th.empty_like(ws_p)
    dqkv = th.empty_like(qkv); dR = [th.empty_like(r) for r in R]
    f = lambda: nat.attn_temporal_bwd(qkv, d_o, R[0], R[1], R[2], mask, ws_p, ws_ds, dqkv, dR[0], dR[1], dR[2], B, T, P, C, heads)
    for _ in range(3): f()
    th.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 20)
    print(f"P={P} C={C}: backward (rows + cols + rpe) {min(ts):.1f} us per call")
