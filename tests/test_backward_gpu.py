"""Training path on the MI355X: gradients of the native U-Net vs the CPU oracle's autograd (which is
pinned to the reference's gradients by tests/golden/backward_*.npz).  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import recipe, unet_oracle as uo
from conftest import GOLDEN
from test_oracle_golden import load_case
from test_forward_gpu import build_native

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["micro", "micro_rb2"])
def test_parameter_gradients_match_oracle(name):
    g = np.load(os.path.join(GOLDEN, f"backward_{name}.npz"))
    cfg, sd, inp = load_case(name)
    model = build_native(cfg, sd).train()
    d = {k: v.cuda() for k, v in inp.items()}
    probe = torch.from_numpy(recipe.gaussianish(name + "/probe", inp["x"].numel()).reshape(inp["x"].shape).astype(np.float32))
    out, _ = model(d["x"], x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"],
                   obs_mask=d["obs_mask"], latent_mask=d["latent_mask"])
    # the differentiable path must agree with the no-grad engine
    with torch.no_grad():
        out_ng, _ = model(d["x"], x0=d["x0"], timesteps=d["t"].float(), frame_indices=d["frame_indices"],
                          obs_mask=d["obs_mask"], latent_mask=d["latent_mask"])
    assert torch.allclose(out.detach(), out_ng, atol=2e-4), float((out.detach() - out_ng).abs().max())
    (out * probe.cuda()).sum().backward()
    # oracle gradients (CPU autograd)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o, _ = uo.unet_forward(sdo, cfg, inp["x"], inp["x0"], inp["t"].float(), inp["frame_indices"], inp["obs_mask"], inp["latent_mask"])
    (o * probe).sum().backward()
    gmax = float(g["gmax"])
    keys = [str(k) for k in g["keys"]]
    worst, worst_key = 0.0, None
    for i, (k, p) in enumerate(model.named_parameters()):
        assert k == keys[i]
        ref = sdo[k].grad
        assert p.grad is not None, k
        err = float((p.grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-3 * gmax)
        if err > worst:
            worst, worst_key = err, k
        # and against the REAL reference's per-tensor gradient norms
        assert abs(float(p.grad.double().norm()) - float(g["norms"][i])) < 3e-3 * (float(g["norms"][i]) + 1e-3 * gmax * np.sqrt(ref.numel())), k
    print(f"[{name}] worst relative gradient error vs oracle: {worst:.2e} ({worst_key})")
    assert worst < 2e-3, (worst, worst_key)
