"""Developer aid: time lfvdm_conv_wgrad on the layer shapes of cfg C (graph replay of 10 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat
dev = th.device("cuda")

def timeit(fn, reps=10, rounds=5):
    fn(); th.cuda.synchronize()
    g = th.cuda.CUDAGraph()
    with th.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / reps)
    return best

shapes = [(40, 16, 128, 0, 128, 3, 1), (40, 16, 128, 128, 128, 3, 1), (40, 8, 256, 0, 256, 3, 1), (40, 8, 256, 256, 256, 3, 1),
          (40, 4, 256, 0, 256, 3, 1), (40, 2, 256, 0, 256, 3, 1), (40, 16, 128, 0, 384, 1, 0), (40, 16, 128, 0, 128, 1, 0),
          (40, 8, 256, 0, 768, 1, 0), (1, 800, 256, 0, 256, 1, 0)]
for (N, H, C0, C1, Cout, k, coef) in shapes:
    coef = coef if os.environ.get("WG_COEF") else 0      # the training plan feeds materialised (raw) operands
    Cin = C0 + C1
    M = N * H * H if k == 3 else N * H * (H if N > 1 else 1)
    Hs, Ws = (H, H) if k == 3 else ((H * H, 1) if N > 1 else (H, 1))
    a = th.randn(N * Hs * Ws, C0, device=dev)
    b = th.randn(N * Hs * Ws, C1, device=dev) if C1 else None
    dout = th.randn(N * Hs * Ws, Cout, device=dev)
    cA = th.rand(N, Cin, device=dev) + 0.5 if coef else None
    cB = th.randn(N, Cin, device=dev) if coef else None
    gw = th.zeros(Cout, Cin, k, k, device=dev); gb = th.zeros(Cout, device=dev)
    kw = dict(src0=a, src1=b, C0=C0, C1=C1, N=N, Hs=Hs, Ws=Ws, Ho=Hs, Wo=Ws, ksize=k, res=dout, ldr=Cout, out=gw, bias=gb, Cout=Cout,
              out_mode=int(os.environ.get("WG_MODE", "1")) if k == 3 else 0)
    if coef:
        kw.update(coefA=cA, coefB=cB, act=nat.ACT_SILU)
    t = timeit(lambda: nat.conv_wgrad(**kw))
    fl = 2.0 * N * Hs * Ws * Cout * Cin * k * k
    print(f"N={N} H={H} C0={C0} C1={C1} Cout={Cout} k={k} coef={coef}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s", flush=True)
