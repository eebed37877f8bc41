"""Developer aid: run every legal tune code of one conv shape several times and print max|d| vs torch.
usage: python tools/conv_codes_check.py N Cin Cout H k stride [repeats]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch, torch.nn.functional as F
from improved_diffusion import _native as nat
N, Cin, Cout, H, k, stride = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
torch.manual_seed(0)
x, w, b = torch.randn(N, Cin, H, H), torch.randn(Cout, Cin, k, k) * 0.05, torch.randn(Cout)
ref = F.conv2d(x, w, b, padding=1 if k == 3 else 0, stride=stride)
Ho = ref.shape[2]
xc = x.permute(0, 2, 3, 1).contiguous().cuda()
wp = torch.empty(Cout, k * k, Cin, device="cuda")
nat.pack_conv_weight(w.cuda().contiguous(), wp)
out = torch.empty(N * Ho * Ho, Cout, device="cuda")
ws = torch.empty(1 << 22, device="cuda"); cnt = torch.zeros(4096, dtype=torch.int32, device="cuda")
bb = b.cuda()
a = nat.fill_conv_args(src0=xc, W=wp, bias=bb, C0=Cin, N=N, Hs=H, Ws=H, Ho=Ho, Wo=Ho, Cout=Cout, out=out, ldo=Cout, ksize=k, stride=stride)
a.splitk_ws, a.splitk_cnt, a.splitk_ws_floats, a.splitk_cnt_ints = ws.data_ptr(), cnt.data_ptr(), ws.numel(), cnt.numel()
codes = (C.c_int * 256)()
n = nat.lib().lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
for code in [codes[i] for i in range(n)]:
    errs = []
    for _ in range(reps):
        a.tune = code
        out.fill_(float("nan"))
        nat.conv_igemm_struct(a)
        got = out.view(N, Ho, Ho, Cout).permute(0, 3, 1, 2).cpu()
        d = (got - ref).abs()
        errs.append(float(d.max()) if torch.isfinite(d).all() else float("nan"))
    t = code - 1
    tag = f"id={t & 15} kch={64 if t & 16 else 32} kz={1 << ((t >> 5) & 7)} gl={((t >> 8) & 3) + 1 if (t >> 8) & 3 else 0}"
    bad = [e for e in errs if not e < 5e-5]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        ev0.record()
        for _ in range(10):
            nat.conv_igemm_struct(a)
        ev1.record(); ev1.synchronize()
        best = min(best, ev0.elapsed_time(ev1) * 100)
    fl = 2.0 * N * Ho * Ho * Cout * Cin * k * k
    print(f"code {code:4d} {tag:32s} {'BAD' if bad else 'ok '} {best:8.1f} us {fl / best / 1e6:6.1f} TF/s  " + " ".join(f"{e:.1e}" for e in errs[:2]), flush=True)
    if bad:
        bn = d.amax(dim=(1,))    # (N, Ho, Ho)
        idx = (d > 5e-5).nonzero()
        print("   bad elements:", idx.shape[0], "first", idx[:4].tolist(), "last", idx[-2:].tolist(), "ticket sum", int(cnt.abs().sum()))
