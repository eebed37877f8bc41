"""Developer aid: per-launch time of the fused GroupNorm backward (lfvdm_gn_bwd_fused / _sums) on the cfg-C training shapes,
50 launches per graph replay (with FiLM + atomics, atomics only, deterministic sums).  usage: python tools/gn_bwd_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat
dev = th.device("cuda"); th.cuda.set_device(0)
L = nat.lib()
N, T = 40, 20
shapes = [(128, 0, 256), (256, 0, 64), (256, 0, 16), (256, 0, 4), (256, 128, 256), (256, 256, 64), (256, 256, 16), (128, 128, 256)]
g = th.Generator(device=dev); g.manual_seed(1)
for C0, C1, P in shapes:
    C = C0 + C1
    r = lambda *s: th.randn(*s, device=dev, generator=g)
    da, a = r(N * P, C), r(N * P, C0)
    b = r(N * P, C1) if C1 else None
    cA, cB = r(N, C), r(N, C)
    stats = th.stack([r(N, 32), r(N, 32).abs() + 0.5], -1).contiguous()
    gamma, beta, film = r(C), r(C), r(N // T, 2 * C)
    dg, db, dfilm = th.zeros(C, device=dev), th.zeros(C, device=dev), th.zeros(N // T, 2 * C, device=dev)
    dxa = th.empty(N * P, C0, device=dev); dxb = th.empty(N * P, C1, device=dev) if C1 else None
    add = r(N * P, C)
    sums = th.empty(N, C, 2, device=dev)
    def atom():
        nat.check(L.lfvdm_gn_bwd_fused(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats), 1,
                                       nat.ptr(dxa), nat.ptr(dxb), nat.ptr(gamma), nat.ptr(beta), film.data_ptr(), film.stride(0), T,
                                       nat.ptr(dg), nat.ptr(db), dfilm.data_ptr(), dfilm.stride(0), nat.ptr(add), add.stride(0),
                                       None, 0, nat.stream()), "fused")
    def nofilm():
        nat.check(L.lfvdm_gn_bwd_fused(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats), 1,
                                       nat.ptr(dxa), nat.ptr(dxb), nat.ptr(gamma), nat.ptr(beta), None, 0, T,
                                       nat.ptr(dg), nat.ptr(db), None, 0, None, 0, None, 0, nat.stream()), "fused")
    def det():
        nat.check(L.lfvdm_gn_bwd_fused_sums(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats), 1,
                                            nat.ptr(dxa), nat.ptr(dxb), nat.ptr(add), add.stride(0), None, 0, nat.ptr(sums), nat.stream()), "sums")
    out = []
    for name, f in (("film+atomics", atom), ("atomics", nofilm), ("sums", det)):
        s = th.cuda.Stream()
        with th.cuda.stream(s):
            for _ in range(3): f()
            s.synchronize()
            gr = th.cuda.CUDAGraph()
            with th.cuda.graph(gr, stream=s):
                for _ in range(50): f()
            gr.replay(); s.synchronize()
            e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10): gr.replay()
            e1.record(s); s.synchronize()
        out.append(f"{name} {e0.elapsed_time(e1) * 1e3 / 500:6.2f} us")
    print(f"C0={C0:4d} C1={C1:4d} P={P:4d}: " + "   ".join(out), flush=True)
