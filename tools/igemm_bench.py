#!/usr/bin/env python3
"""Developer aid: time every implicit-GEMM launch code of one layer shape (HIP events).
usage: python3 tools/igemm_bench.py N H Cin Cout [k]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat

N, H, Cin, Cout = (int(v) for v in sys.argv[1:5])
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = th.device("cuda", 0)
x = th.randn(N * H * H, Cin, device=dev)
w = th.randn(Cout, k * k, Cin, device=dev) * 0.02
b = th.randn(Cout, device=dev)
out = th.empty(N * H * H, Cout, device=dev)
a = nat.fill_conv_args(src0=x, C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, W=w, bias=b, Cout=Cout, out=out, ldo=Cout)
L, s = nat.lib(), nat.stream()
codes = (C.c_int * 256)()
n = L.lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
flops = 2.0 * N * H * H * Cout * k * k * Cin
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
res = []
for code in [0] + [codes[i] for i in range(n)]:
    a.tune = code
    if L.lfvdm_conv_igemm(C.byref(a), s) != 0:
        continue
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(6):
            L.lfvdm_conv_igemm(C.byref(a), s)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6)
    t = code - 1
    res.append((best, code, t & 15, 64 if t & 16 else 32, (t >> 5) & 7, ((t >> 8) & 3) + 1))
for best, code, cid, kch, kzl, gl in sorted(res)[:10]:
    print(f"code {code:5d} cfg {cid:2d} kch {kch} kz-index {kzl} stages {gl}: {best * 1e3:8.1f} us  {flops / best / 1e9:7.1f} TFLOP/s")
for best, code, cid, kch, kzl, gl in sorted(res):
    if cid >= 8:
        print(f"  new: code {code:5d} cfg {cid:2d} kch {kch} kz-index {kzl} stages {gl}: {best * 1e3:8.1f} us  {flops / best / 1e9:7.1f} TFLOP/s")
