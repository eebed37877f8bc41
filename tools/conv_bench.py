"""Developer aid: steady-state throughput of the conv kernel on large problems + per-config sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
from improved_diffusion import _native as nat

def run(N, Cin, Cout, H, k=3, coef=True, reps=20):
    x = th.randn(N * H * H, Cin, device="cuda")
    w = th.randn(Cout, k * k, Cin, device="cuda") * 0.05
    b = th.randn(Cout, device="cuda")
    cA = th.randn(N, Cin, device="cuda"); cB = th.randn(N, Cin, device="cuda")
    out = th.empty(N * H * H, Cout, device="cuda")
    kw = dict(src0=x, C0=Cin, N=N, Hs=H, Ws=H, Ho=H, Wo=H, ksize=k, W=w, bias=b, Cout=Cout, out=out, ldo=Cout)
    # (coef: the operand-prologue variant of rounds 1-2 is gone; the argument is kept so that old command lines still run)
    for _ in range(3):
        nat.conv_igemm(**kw)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        nat.conv_igemm(**kw)
    e1.record(); th.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    return us, fl / us / 1e6

if __name__ == "__main__":
    a = th.randn(4096, 4096, device="cuda"); b = th.randn(4096, 4096, device="cuda")
    for _ in range(3): a @ b
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): a @ b
    th.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"torch fp32 matmul 4096^3: {2*4096**3/dt/1e12:.1f} TF/s")
    cfgs = os.environ.get("CFGS", "-1").split(",")
    for (N, Cin, Cout, H, k) in [(40, 64, 64, 16, 3), (2560, 64, 64, 16, 3), (2560, 128, 128, 16, 3), (40, 128, 128, 2, 3), (40, 128, 128, 8, 3), (40, 64, 192, 16, 1)]:
        us, tf = run(N, Cin, Cout, H, k)
        us2, tf2 = run(N, Cin, Cout, H, k, coef=False)
        print(f"N={N} Cin={Cin} Cout={Cout} H={H} k={k}: {us:9.1f} us {tf:6.1f} TF/s | no-prologue {us2:9.1f} us {tf2:6.1f} TF/s")
