"""Developer aid: fused spatial attention (qkv projection inside) against the two-launch form, per cfg-B shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat

def timeit(fn, reps=30):
    for _ in range(5): fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps

L = nat.lib()
for (N, P, C, h) in [(40, 256, 64, 4), (40, 64, 128, 4), (40, 4, 128, 4), (28, 256, 64, 4)]:
    M = N * P
    xn = th.randn(M, C, device="cuda"); W = th.randn(3 * C, C, device="cuda") * 0.1; b = th.randn(3 * C, device="cuda")
    o = th.empty(M, C, device="cuda"); qkv = th.empty(M, 3 * C, device="cuda"); o2 = th.empty(M, C, device="cuda")
    s = nat.stream()
    f = lambda: L.lfvdm_attn_spatial_fused(xn.data_ptr(), W.data_ptr(), b.data_ptr(), o.data_ptr(), N, P, C, h, s)
    def u():
        nat.conv_igemm(src0=xn, C0=C, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=b, Cout=3 * C, out=qkv, ldo=3 * C)
        nat.attn_spatial(qkv, o2, None, N, P, C, h)
    tf, tu = timeit(f), timeit(u)
    print(f"N={N} P={P} C={C} heads={h}: fused {tf:6.1f} us   qkv + attention {tu:6.1f} us   max|d| {float((o - o2).abs().max()):.2e}  waves={os.environ.get('LFVDM_SF_WAVES', 'auto')}")
