"""Developer aid: lfvdm_proj_gn against lfvdm_conv_igemm (1x1, residual) + lfvdm_gn_apply on 16x16 frames (50 launches per
graph replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat
L = nat.lib()
dev = th.device("cuda")

def graph_time(f, n=50, reps=10):
    s = th.cuda.Stream()
    with th.cuda.stream(s):
        for _ in range(3): f()
        s.synchronize()
        g = th.cuda.CUDAGraph()
        with th.cuda.graph(g, stream=s):
            for _ in range(n): f()
        g.replay(); s.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)

for (N, C, P) in [(40, 64, 256), (20, 64, 256), (160, 64, 256), (40, 128, 256), (40, 128, 64), (20, 128, 64), (160, 128, 64)]:
    M = N * P
    o = th.randn(M, C, device=dev); res = th.randn(M, C, device=dev); W = th.randn(C, C, device=dev) * 0.1; bias = th.randn(C, device=dev)
    gam = th.randn(C, device=dev); bet = th.randn(C, device=dev)
    out = th.empty(M, C, device=dev); y = th.empty(M, C, device=dev); yn = th.empty(M, C, device=dev)
    def fused():
        nat.check(L.lfvdm_proj_gn(o.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5,
                                  nat.ACT_NONE, out.data_ptr(), None, N, P, C, nat.stream()), "proj_gn")
    def two():
        nat.conv_igemm(src0=o, C0=C, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=bias, Cout=C, res=res, ldr=C, out=y, ldo=C)
        nat.check(L.lfvdm_gn_apply(y.data_ptr(), None, C, 0, N, P, gam.data_ptr(), bet.data_ptr(), None, 1, 0, 1e-5, nat.ACT_NONE,
                                   yn.data_ptr(), None, None, None, nat.stream()), "gn_apply")
    tf, tt = graph_time(fused), graph_time(two)
    print(f"N={N:4d} C={C:4d} P={P:4d}: fused {tf:6.2f} us   1x1 GEMM + gn_apply {tt:6.2f} us   max|d| {float((out - yn).abs().max()):.2e}", flush=True)
