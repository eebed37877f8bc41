"""Developer aid: lfvdm_gn_temporal_qkv against lfvdm_gn_temporal + the 1x1 lfvdm_conv_igemm, per cfg-B / cfg-C shape
(50 launches per graph replay).  LFVDM_TQ_SPLIT forces the number of column shares."""
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

for (B, T, P, C) in [(2, 20, 256, 64), (2, 20, 64, 128), (2, 20, 16, 128), (2, 20, 4, 128), (1, 20, 256, 64), (8, 20, 256, 64), (2, 20, 256, 128), (2, 20, 64, 256)]:
    if L.lfvdm_gn_temporal_qkv_ok(B, T, P, C) != 0:
        continue
    M = B * T * P
    x = th.randn(M, C, device=dev); gam = th.randn(C, device=dev); bet = th.randn(C, device=dev)
    W = th.randn(3 * C, C, device=dev) * 0.1; bias = th.randn(3 * C, device=dev)
    xn = th.empty(M, C, device=dev); qkv = th.empty(M, 3 * C, device=dev); qkv2 = th.empty(M, 3 * C, device=dev); xn2 = th.empty(M, C, device=dev)
    def fused():
        nat.check(L.lfvdm_gn_temporal_qkv(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5, xn.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                          qkv.data_ptr(), B, T, P, C, nat.stream()), "tq")
    a = nat.conv_args(src0=xn2, C0=C, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=bias, Cout=3 * C, out=qkv2, ldo=3 * C) if hasattr(nat, "conv_args") else None
    def two():
        nat.gn_temporal(x, gam, bet, 1e-5, xn2, B, T, P, C)
        nat.conv_igemm(src0=xn2, C0=C, N=B * T, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=W, bias=bias, Cout=3 * C, out=qkv2, ldo=3 * C)
    tf, tt = graph_time(fused), graph_time(two)
    print(f"B={B} T={T} P={P:4d} C={C:4d}: fused {tf:6.2f} us   gn_temporal + qkv GEMM {tt:6.2f} us   max|d| {float((qkv - qkv2).abs().max()):.2e}  split={os.environ.get('LFVDM_TQ_SPLIT', 'auto')}", flush=True)
