import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import run
N, Cin, Cout, H, k = [int(v) for v in sys.argv[1:6]]
coef = len(sys.argv) > 6 and sys.argv[6] == "1"
us, tf = run(N, Cin, Cout, H, k, coef=coef, reps=50)
print(f"N={N} Cin={Cin} Cout={Cout} H={H} k={k} coef={coef}: {us:.1f} us {tf:.1f} TF/s")
