import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from improved_diffusion import _native as nat
from test_ops_gpu import _temporal_core_f64, rnd
from oracle import recipe
B, T, P, Cc, heads = 2, 20, 256, 64, 4
M = B * T * P
qkv = rnd("t2/qkv", M, 3 * Cc)
Rs = [0.3 * rnd(f"t2/R{i}", B, T, T, Cc) for i in range(3)]
mask = (torch.from_numpy(recipe.uniform_pm1("t2/mask", B * T)).view(B, T) > 0).float()
ref = _temporal_core_f64(qkv.double(), Rs[0].double(), Rs[1].double(), Rs[2].double(), mask.double(), B, T, P, Cc, heads).float()
g = [t.cuda().contiguous() for t in (qkv, *Rs, mask)]
L = nat.lib()
for cfg in [(0, 0), (4, 4), (4, 1), (2, 4), (4, 2)]:
    L.lfvdm_attn_temporal2_debug(cfg[0], cfg[1], 0)
    for rep in range(2):
        o = torch.full((M, Cc), float("nan"), device="cuda")
        nat.attn_temporal(g[0], g[1], g[2], g[3], g[4], o, None, B, T, P, Cc, heads)
        d = (o.cpu() - ref).abs().view(B, T, P, heads, Cc // heads).amax(-1)     # (B, T, P, heads)
        bad = (d > 1e-4) | torch.isnan(d)
        print(cfg, rep, "bad (b,t,p,h):", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(d).sum()))
        if bad.any():
            idx = bad.nonzero()
            print("   b:", sorted(set(idx[:, 0].tolist())), "t:", sorted(set(idx[:, 1].tolist())), "h:", sorted(set(idx[:, 3].tolist())),
                  "p:", sorted(set(idx[:, 2].tolist()))[:40])
