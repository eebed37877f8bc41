#!/usr/bin/env python3
"""Developer aid: time every weight-gradient launch code of one layer shape (HIP events, scratch outputs).
usage: python3 tools/wgrad_bench.py N H Cin Cout [k]"""
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
d = th.randn(N * H * H, Cout, device=dev)
gp = th.zeros(Cout, k * k, Cin, device=dev)
db = th.zeros(Cout, device=dev)
a = nat.fill_conv_args(src0=x, C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, res=d, ldr=Cout, out=gp, bias=db, Cout=Cout)
L, s = nat.lib(), nat.stream()
flops = 2.0 * N * H * H * Cout * k * k * Cin
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
res = []
for code in [0] + nat._wgrad_codes(a):
    a.tune = code
    if L.lfvdm_conv_wgrad(C.byref(a), s) != 0:
        continue
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(10):
            L.lfvdm_conv_wgrad(C.byref(a), s)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    t = code - 1
    res.append((best, code, t & 3, (t >> 2) & 3, t >> 4))
for best, code, tile, st, ms in sorted(res)[:12]:
    print(f"code {code:5d} tile {tile} stages {st} m-slices {ms:4d}: {best * 1e3:8.1f} us  {flops / best / 1e9:7.1f} TFLOP/s")
for best, code, tile, st, ms in sorted(res):
    if st == 0 and code:
        print(f"  taps: code {code:5d} tile {tile} m-slices {ms:4d}: {best * 1e3:8.1f} us  {flops / best / 1e9:7.1f} TFLOP/s")
