"""Developer aid: in-kernel phase stamps (clock64) of workgroup (0,0) for one conv launch, from a library built
with -DLFVDM_STAMP (devlib/liblfvdm_stamp.so).  usage: tools/conv_stamps.py N Cin Cout H k coef [cfg] [kch64]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
os.environ["LFVDM_AUTOTUNE"] = "0"
import torch as th
from improved_diffusion import _native as nat
nat.LIB_PATH = os.path.join(ROOT, "devlib", os.environ.get("STAMP_LIB", "liblfvdm_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import run
N, Cin, Cout, H, k, coef = [int(v) for v in sys.argv[1:7]]
for cfg in (sys.argv[7].split(",") if len(sys.argv) > 7 else ["-1"]):
    # note: LFVDM_CONV_CFG is read once per process (static) -> one cfg per process run
    us, tf = run(N, Cin, Cout, H, k, coef=bool(coef), reps=20)
    st = (C.c_ulonglong * 128)()
    nat.lib().lfvdm_debug_stamps.argtypes = [C.c_void_p]
    nat.lib().lfvdm_debug_stamps(st)
    d = [st[i + 1] - st[i] for i in range(3)]
    print(f"cfg={os.environ.get('LFVDM_CONV_CFG','model')} {us:.1f} us {tf:.1f} TF/s | cycles: prologue {d[0]} loop {d[1]} reduce+epilogue {d[2]} total {st[3]-st[0]} (100 MHz ticks? x24 for 2.4GHz)")
