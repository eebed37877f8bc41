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


def _worker8(rank, world, port, q):
    """World size 8 on gloo: what the 8-GPU node runs, minus the kernels."""
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LFVDM_AUTOTUNE="0", OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    import time
    from improved_diffusion import dist_util, _native as nat
    from improved_diffusion._exchange import GradExchange, plan_buckets
    from improved_diffusion.train_util import ParamArena
    from improved_diffusion.unet import UNetVideoModel
    dist_util.setup_dist()
    assert dist.get_backend() == "gloo" and dist.get_world_size() == world == 8
    torch.manual_seed(100 + rank)
    model = UNetVideoModel(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=(1,),
                           channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True, use_rpe_net=True)
    named = list(model.named_parameters())
    groups, marks = plan_buckets(named, 5)                   # the default bucket count of the training loop
    arena = ParamArena([p for _, p in named], groups)
    nb = len(arena.bucket_ranges)                            # (this small model has stages for 4 of the 5 buckets)
    assert 3 <= nb <= 5 and arena.bucket_ranges[-1][1] == arena.numel and len(marks) == nb - 1
    xch = GradExchange(arena, marks)
    assert xch.world == 8
    xch.broadcast(arena.p)
    # 1. mean divisor: rank r contributes (r + 1) * pattern, the mean over 8 ranks is 4.5 * pattern
    arena.zero_grad()
    for i, p in enumerate(model.parameters()):
        p.grad.add_(float(rank + 1) * (i % 5 + 1))
    # 2. the skip word rides in the LAST bucket's SUM: raised (1.0) on ranks 2, 5 and 6 only, every rank must see 3.0 - and
    #    3 / 8 != 0 after the optimizer's 1 / world scale: every replica skips
    w = xch.skip_word()
    assert w is not None and float(w) == 0.0
    if rank in (2, 5, 6):
        w.fill_(1.0)
    xch.launch()
    xch.wait()
    assert float(xch.skip_word()) == 3.0 and float(xch.skip_word()) / world != 0.0
    mean = arena.g / world
    for i, v in enumerate(arena.views(mean)):
        assert torch.allclose(v, torch.full_like(v, 4.5 * (i % 5 + 1)))
    arena.zero_grad()
    assert float(xch.skip_word()) == 0.0                     # cleared with the gradients
    # 3. decisions that change the schedule are MIN-agreed: one rank's probe "fails" -> nobody overlaps
    assert GradExchange.agree(rank != 3, torch.device("cpu")) is False
    assert GradExchange.agree(True, torch.device("cpu")) is True
    # 4. weight-gradient launch codes: rank 0 decides (here: a cached code only rank 0 has), 7 ranks wait on the rendezvous
    #    store and read it - no collective, so a rank that resolves the shape late cannot mismatch one
    a = nat.ConvArgs()
    a.C0, a.N, a.Hs, a.Ws, a.Ho, a.Wo, a.ksize, a.stride, a.Cout = 128, 40, 16, 16, 16, 16, 3, 1, 128
    key = nat._wgrad_key(a)
    if rank == 0:
        time.sleep(1.0)                                      # the waiters are parked in store.wait by now
        nat.tune_cache()[key] = 1 + 2 + 4 * 2 + 16 * 7
    t0 = time.time()
    code = nat._tuned_wgrad_code(a, 1)
    assert code == 1 + 2 + 4 * 2 + 16 * 7, (rank, code)
    assert rank == 0 or time.time() - t0 > 0.3               # (they did wait for rank 0)
    assert nat._tuned_wgrad_code(a, 1) == code               # second resolution: from the process-local table
    dist.barrier()
    q.put((rank, code))
    dist.destroy_process_group()


def test_eight_rank_gloo_exchange():
    """The exchange at the world size of the target node: bucket plan (one marker per early bucket), the mean divisor, the skip
    word riding in the last bucket (k / 8 != 0 on every rank), the MIN-agreed overlap decision with one failing rank, and
    rank 0's weight-gradient launch code reaching 7 waiters through the rendezvous store (train_util.py:116-125,
    dist_util.py:21-50 of the reference stand behind DistributedDataParallel for this)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert [r for r, _ in res] == list(range(8)) and len({c for _, c in res}) == 1
