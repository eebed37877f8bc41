"""World-size-2 gloo test of the data-parallel plumbing (runs on CPU): the flat gradient arena is summed
with ONE all-reduce and scaled by 1/world, parameters are broadcast from rank 0, state-dict views stay
intact.  The fused optimizer kernel itself is GPU-only and covered by tests/test_train_gpu.py."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from improved_diffusion import dist_util
    from improved_diffusion.train_util import ParamArena
    from improved_diffusion.unet import UNetVideoModel
    dist_util.setup_dist()
    assert dist.get_backend() == "gloo" and dist.get_world_size() == world and dist_util.dev().type == "cpu"
    torch.manual_seed(100 + rank)   # different replicas on purpose
    model = UNetVideoModel(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=(1,),
                           channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True, use_rpe_net=True)
    keys = list(model.state_dict().keys())
    arena = ParamArena(list(model.parameters()))
    assert list(model.state_dict().keys()) == keys
    assert all(p.data_ptr() == arena.p.data_ptr() + 4 * o for p, o in zip(model.parameters(), arena.offsets))
    dist.broadcast(arena.p, 0)                                   # initial replica sync (TrainLoop.__init__)
    chk = arena.p.double().sum()
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    assert all(torch.equal(allc[0], c) for c in allc)
    # rank-dependent gradients written through the parameter views, one collective, then the mean
    arena.zero_grad()
    for i, p in enumerate(model.parameters()):
        p.grad.add_(float(rank + 1) * (i % 7 + 1))
    dist.all_reduce(arena.g, op=dist.ReduceOp.SUM)
    mean = arena.g / world
    for i, (p, v) in enumerate(zip(model.parameters(), arena.views(mean))):
        want = sum(r + 1 for r in range(world)) / world * (i % 7 + 1)
        assert torch.allclose(v, torch.full_like(v, want))
    # sync_params helper (reference dist_util.py:66-72)
    t = torch.full((3,), float(rank))
    dist_util.sync_params([t])
    assert torch.equal(t, torch.zeros(3))
    dist.barrier()
    q.put((rank, float(chk)))
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_arena():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res[0][1] == res[1][1]
