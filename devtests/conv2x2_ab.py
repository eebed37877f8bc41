"""Developer A/B: a 3x3 convolution on 2x2 maps vs the equivalent Linear(4*Cin -> 4*Cout) on the [n][pixel][c] rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch as th
from improved_diffusion import _native as nat
from conv_bench import run
for N in (40,):
    for Cin, Cout in ((256, 256), (512, 256), (192, 256)):
        a, _ = run(N, Cin, Cout, 2, 3, coef=False, reps=200)
        b, _ = run(N, 4 * Cin, 4 * Cout, 1, 1, coef=False, reps=200)
        print(f"N={N} {Cin}->{Cout}: 3x3 on 2x2 {a:.2f} us | linear {4*Cin}->{4*Cout} on 1x1 {b:.2f} us")
for Cin, Cout in ((192, 192), (384, 192), (256, 192)):
    a, _ = run(40, Cin, Cout, 4, 3, coef=False, reps=200)
    print(f"4x4: N=40 {Cin}->{Cout}: 3x3 {a:.2f} us")
