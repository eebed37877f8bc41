"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``latent-flexible-video-diffusion-modeling_amd/``) and the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Two GPU tests create a world-1 process group INSIDE the pytest process (``dist_util.setup_dist()`` as the reference's
    scripts do, scripts/video_train.py:93): shut it down before the interpreter exits (torch warns about a leaked
    ProcessGroupNCCL otherwise)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass
