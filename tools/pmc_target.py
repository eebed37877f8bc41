"""Target program for rocprofv3 counter passes (developer aid): builds the autotuned cfg-B sampler exactly as
bench.py does, then runs a few denoising steps EAGERLY (one dispatch per kernel, visible to the profiler)
bracketed by two q_sample launches that serve as markers in the dispatch list.

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_target.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_target.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 tools/pmc_target.py
    python3 tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json gpurun_out/pmc_mfma
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))
import torch as th
import bench
from improved_diffusion import _native as nat

dev = th.device("cuda")
th.cuda.set_device(0)
model, diffusion = bench.make_model_and_diffusion(64, dev)
B, T = 2, 20
shape = (B, T, 4, 16, 16)
inputs = bench.synthetic_inputs(B, T, 0, dev)
sampler = diffusion._graph_sampler(model, shape, True)
th.manual_seed(1234)
sampler.begin(th.randn(*shape, device=dev), inputs)
th.cuda.synchronize()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = th.randn(*shape, device=dev); out = th.empty_like(x)
tb = diffusion.tables(dev)
t = th.full((B,), 500, device=dev, dtype=th.int64)
def marker():
    nat.q_sample(x, x, t, tb["sqrt_alphas_cumprod"], tb["sqrt_one_minus_alphas_cumprod"], out)
with th.no_grad():
    marker()
    for _ in range(steps):
        sampler._step_body()
    marker()
th.cuda.synchronize()
print("eager steps:", steps, "plan launches per step:", len(sampler.plan.steps))
