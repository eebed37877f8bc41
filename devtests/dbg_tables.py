import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, time
from test_oracle_golden import load_case
from test_forward_gpu import build_native
from test_sampler_gpu import make_diffusion
from improved_diffusion._engine import Plan
cfg, sd, inp = load_case("micro")
model = build_native(cfg, sd)
diff = make_diffusion(1000, "250")
d = {k: v.cuda() for k, v in inp.items()}
mk = dict(frame_indices=d["frame_indices"], obs_mask=d["obs_mask"], latent_mask=d["latent_mask"], x0=d["x0"])
shape = tuple(inp["x"].shape)
s = diff._graph_sampler(model, shape, True)
s.begin(d["x"].clone(), mk)
pl = s.plan
print("time_steps", pl.time_steps, "build ms", s.table_build_ms)
# per-step plan for comparison
eng = model.native_engine()
p2 = Plan(eng, shape[0], shape[1], shape[3], shape[4], False)
p2.refresh_weights()
for i in (249, 100, 0):
    ts = s.ts_table[i].item()
    p2.set_inputs(d["x"], d["x0"], torch.full((shape[0],), ts, device="cuda"), d["frame_indices"], d["obs_mask"], d["latent_mask"])
    p2.launch(); torch.cuda.synchronize()
    B = shape[0]
    ra = pl.rows_all.view(pl.time_steps, B, -1)[i]
    print(i, "rows diff", float((ra - p2.rows).abs().max()), "film part", float((ra[:, :pl.film_floats] - p2.rows[:, :pl.film_floats]).abs().max()))
    for (r, Ra), (r2, Rb) in zip(pl.R.items(), p2.R.items()):
        dd = float((Ra.view(pl.time_steps, B, *Ra.shape[1:])[i] - Rb).abs().max())
        if dd > 0: print("  R diff", dd)
# timing of the builders
for name, fn in (("time tables", lambda: pl.build_time_tables(s.ts_table)), ("R tables", lambda: pl.build_R_tables(d["frame_indices"]))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) * 1e3, "ms")
# cfg B sizes
import bench
m2, df2 = bench.make_model_and_diffusion(64, torch.device("cuda"))
inp2 = bench.synthetic_inputs(2, 20, 0, torch.device("cuda"))
s2 = df2._graph_sampler(m2, (2, 20, 4, 16, 16), True)
os.environ["LFVDM_AUTOTUNE"] = "0"
s2.begin(torch.randn(2, 20, 4, 16, 16, device="cuda"), inp2)
for name, fn in (("cfgB time tables", lambda: s2.plan.build_time_tables(s2.ts_table)), ("cfgB R tables", lambda: s2.plan.build_R_tables(inp2["frame_indices"]))):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) * 1e3, "ms")
