"""World-size-2 gloo test of the data-parallel plumbing (runs on CPU) through the PRODUCT's exchange code: the
gradient arena laid out in backward-order buckets (``plan_buckets`` + ``ParamArena``), the parameter broadcast and the
bucketed SUM all-reduce of ``_exchange.GradExchange`` (what ``TrainLoop.__init__`` / ``optimize_normal`` call), scaled
by 1/world.  The fused optimizer kernel and the overlap with a replayed backward graph are GPU-only: they run at
world size 2 in tests/test_dist_gpu.py."""
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
    from improved_diffusion._exchange import GradExchange, plan_buckets, is_late
    from improved_diffusion.train_util import ParamArena
    from improved_diffusion.unet import UNetVideoModel
    dist_util.setup_dist()
    assert dist.get_backend() == "gloo" and dist.get_world_size() == world and dist_util.dev().type == "cpu"
    torch.manual_seed(100 + rank)   # different replicas on purpose
    model = UNetVideoModel(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=(1,),
                           channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True, use_rpe_net=True)
    keys = list(model.state_dict().keys())
    named = list(model.named_parameters())
    groups, marks = plan_buckets(named, 3)
    arena = ParamArena([p for _, p in named], groups)
    # layout: buckets are contiguous, ordered as the backward pass finishes them; late parameters in the last one
    assert len(arena.bucket_ranges) == 3 and arena.bucket_ranges[0][0] == 0 and arena.bucket_ranges[-1][1] == arena.numel
    assert all(a[1] == b[0] for a, b in zip(arena.bucket_ranges, arena.bucket_ranges[1:]))
    late = [i for i, (n, _) in enumerate(named) if is_late(n)]
    assert late and set(late) <= set(groups[-1])
    assert any(n.startswith("out.") for n, _ in (named[i] for i in groups[0])), "the head finishes first in the backward pass"
    assert len(marks) == 2 and sorted(marks.values()) == [0, 1]
    assert list(model.state_dict().keys()) == keys
    assert all(p.data_ptr() == arena.p.data_ptr() + 4 * o for p, o in zip(model.parameters(), arena.offsets))
    xch = GradExchange(arena, marks)
    assert xch.world == world and not xch.overlap            # no streams on CPU tensors
    xch.broadcast(arena.p)                                   # initial replica sync (TrainLoop.__init__)
    chk = arena.p.double().sum()
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    assert all(torch.equal(allc[0], c) for c in allc)
    # rank-dependent gradients written through the parameter views, one collective per bucket, then the mean
    arena.zero_grad()
    for i, p in enumerate(model.parameters()):
        p.grad.add_(float(rank + 1) * (i % 7 + 1))
    xch.launch()
    xch.wait()
    assert xch.stats["exchanges"] == 1
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


def test_two_rank_gloo_gradient_exchange():
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
