"""``TrainLoop`` at world size 2 on the MI355X: two fresh processes share cuda:0 (gloo backend - RCCL refuses two ranks
on one device; everything above the collective call is the code the 8-GPU job runs).  Each rank trains on different data;
after ``run_step`` both ranks must hold identical parameters, equal to a single-process AdamW/EMA step on the MEAN of the
two ranks' gradients (reference train_util.py:116-125 DDP wrap + :346-357 optimizer).  Steps 3+ run the captured
micro-step, where the "bucket complete" signals are counter-bumping kernel nodes of the replayed graph.  GPU only."""
import argparse
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(B, T, C, H, seed):
    g = torch.Generator().manual_seed(seed)
    while True:
        yield (torch.randn(B, T, C, H, H, generator=g).clamp(-1, 1), {})


def _worker(rank, world, port, q, microbatch):
    try:
        for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), LFVDM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        from improved_diffusion import dist_util, script_util as su
        from improved_diffusion.train_util import TrainLoop
        from test_oracle_golden import load_case
        from test_forward_gpu import build_native
        dist_util.setup_dist()
        assert dist.get_world_size() == world and dist.get_backend() == "gloo" and dist_util.dev().type == "cuda"
        cfg, sd, _ = load_case("micro")
        if rank == 1:       # a different replica on purpose: the constructor must broadcast rank 0's
            sd = {k: v + 0.01 for k, v in sd.items()}
        model = build_native(cfg, sd).train()
        diffusion = su.create_gaussian_diffusion(steps=1000, rescale_timesteps=True, rescale_learned_sigmas=True)
        loop = TrainLoop(model=model, diffusion=diffusion, data=_data(2, 12, 4, 16, 50 + rank), batch_size=2, microbatch=microbatch,
                         lr=1e-3, ema_rate="0.9", log_interval=1000, save_interval=10 ** 9, resume_checkpoint="", use_fp16=False,
                         diffusion_space_kwargs={}, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.01,
                         lr_anneal_steps=0, sample_interval=None, pad_with_random_frames=True, max_frames=4,
                         enc_dec_chunk_size=20, args=argparse.Namespace(resume_id=""))
        n_buckets = len(loop.arena.bucket_ranges)         # up to LFVDM_GRAD_BUCKETS = 5; a two-level U-Net may give fewer
        assert loop.world == world and loop.use_ddp and 2 <= n_buckets <= 5 and len(loop.exchange.marks) == n_buckets - 1
        # the last bucket holds only what completes at the very end of the backward pass (late parameters + input conv)
        from improved_diffusion._exchange import is_late, stage_of
        names = [n for n, _ in model.named_parameters()]
        assert all(is_late(names[i]) or stage_of(names[i]) in (None, (0, 0)) for i in loop.arena.groups[-1])
        params = list(model.parameters())

        def gathered(t):
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t.contiguous())
            return out

        def flat(tensors):
            return torch.cat([t.detach().reshape(-1) for t in tensors])

        p0 = gathered(flat(params))
        assert torch.equal(p0[0], p0[1]), "parameters must be rank 0's after construction"
        assert torch.equal(flat(loop.ema_params[0]), flat(params))
        torch.manual_seed(100 + rank); np.random.seed(100 + rank)
        # ---- step 1 (eager micro-steps): exchange + fused optimizer against torch AdamW on the mean gradient
        loop.forward_backward()
        local = flat([p.grad for p in params])
        both = gathered(local)
        assert not torch.allclose(both[0], both[1]), "ranks must see different data"
        mean_grad = (both[0] + both[1]) / world
        ref_params = [torch.nn.Parameter(p.detach().clone()) for p in params]
        off = 0
        for rp in ref_params:
            rp.grad = mean_grad[off:off + rp.numel()].view_as(rp).clone()
            off += rp.numel()
        opt = torch.optim.AdamW(ref_params, lr=1e-3, weight_decay=0.01)
        opt.step()
        ref_ema = [p.detach().clone().mul_(0.9).add_(rp.detach(), alpha=0.1) for p, rp in zip(params, ref_params)]
        loop.optimize_normal()
        loop.step += 1
        torch.cuda.synchronize()
        for p, rp in zip(params, ref_params):
            assert torch.allclose(p.detach(), rp.detach(), atol=1e-6, rtol=1e-5)
        for e, re_ in zip(loop.ema_params[0], ref_ema):
            assert torch.allclose(e, re_, atol=1e-6, rtol=1e-5)
        gn_ref = float(mean_grad.double().norm())
        assert abs(float(np.sqrt(loop.grad_sqsum.item())) - gn_ref) < 1e-3 * gn_ref, "grad norm is that of the MEAN gradient"
        # ---- steps 2..6: the micro-step becomes a replayed graph; replicas must stay bit-identical
        for _ in range(5):
            loop.run_step()
            loop.step += 1
        torch.cuda.synchronize()
        assert loop._graph_state.get("graph") is not None, "micro-step was not captured"
        pn = gathered(flat(params))
        assert torch.equal(pn[0], pn[1]), float((pn[0] - pn[1]).abs().max())
        en = gathered(flat(loop.ema_params[0]))
        assert torch.equal(en[0], en[1])
        assert bool(torch.isfinite(pn[0]).all()) and float((pn[0] - p0[0]).abs().max()) > 1e-3
        st = dict(loop.exchange.stats)
        assert st["exchanges"] == 6
        probe_fail = os.environ.get("LFVDM_TEST_PROBE_FAIL_RANK", "")
        if probe_fail:      # ONE rank's overlap probe failed: the decision is collective, nobody overlaps, results as above
            assert not loop.exchange.overlap and loop.exchange.flags is None
            assert loop.exchange.overlap_probe["ok_on_every_rank"] is False
            assert st["buckets_behind_event"] == 0 and st["buckets_behind_graph_end"] == n_buckets * 6, st
        else:
            assert loop.exchange.overlap, "the overlapped exchange is the default on a GPU"
        if loop.exchange.overlap:       # the early buckets behind their counters, the last one behind the end of the graph
            assert st["buckets_behind_event"] == (n_buckets - 1) * 6 and st["buckets_behind_graph_end"] == 6, st
            assert not loop.exchange.flags.timed_out()
        loop.exchange.collect_timing()
        dist.barrier()
        q.put((rank, "ok", st, bool(loop.exchange.overlap), [round(x, 3) for x in loop.exchange.exposed_ms]))
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None))
        raise


@pytest.mark.parametrize("microbatch,deterministic,probe_fail", [(-1, "0", ""), (1, "0", ""), (1, "1", ""), (-1, "0", "1")],
                         ids=["one_microbatch", "two_microbatches", "two_microbatches_deterministic", "one_ranks_probe_fails"])
def test_trainloop_world2_matches_mean_gradient_step(microbatch, deterministic, probe_fail, monkeypatch):
    # deterministic = "1": the same job with LFVDM_DETERMINISTIC=1 (ordered slabs instead of float atomics) in both ranks
    # probe_fail = "1": rank 1's overlap probe is made to fail - every rank must then exchange behind the graph's end (the
    # probe's outcome is agreed with a MIN all-reduce at construction: the schedule of collectives is rank-invariant)
    monkeypatch.setenv("LFVDM_DETERMINISTIC", deterministic)
    monkeypatch.setenv("LFVDM_TEST_PROBE_FAIL_RANK", probe_fail)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, microbatch)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        res.append(q.get(timeout=420))
    for p in procs:
        p.join(60)
    for r in sorted(res):
        print(r)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    assert all(p.exitcode == 0 for p in procs)


def _rccl_worker(port, q):
    """One rank, REAL backend (nccl = RCCL): the exchange is forced on (world pretended to be 2, so every bucket goes
    through dist.all_reduce on RCCL's stream behind its device-side counter; LFVDM_FORCE_EXCHANGE) - the mechanics the
    8-GPU job relies on, minus the wire."""
    try:
        for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        os.environ.pop("LFVDM_DIST_BACKEND", None)
        os.environ["LFVDM_FORCE_EXCHANGE"] = "1"       # the bucketed exchange at world size 1: SUM over one rank = identity
        import torch.distributed as dist
        from improved_diffusion import dist_util, script_util as su
        from improved_diffusion.train_util import TrainLoop
        from test_oracle_golden import load_case
        from test_forward_gpu import build_native
        dist_util.setup_dist()
        assert dist.get_backend() == "nccl"
        cfg, sd, _ = load_case("micro")
        model = build_native(cfg, sd).train()
        diffusion = su.create_gaussian_diffusion(steps=1000, rescale_timesteps=True, rescale_learned_sigmas=True)
        loop = TrainLoop(model=model, diffusion=diffusion, data=_data(2, 12, 4, 16, 7), batch_size=2, microbatch=-1,
                         lr=1e-3, ema_rate="0.9", log_interval=1000, save_interval=10 ** 9, resume_checkpoint="", use_fp16=False,
                         diffusion_space_kwargs={}, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.01,
                         lr_anneal_steps=0, sample_interval=None, pad_with_random_frames=True, max_frames=4,
                         enc_dec_chunk_size=20, args=argparse.Namespace(resume_id=""))
        assert loop.use_ddp and loop.world == 1 and loop.exchange.world == 2 and len(loop.arena.bucket_ranges) >= 2
        torch.manual_seed(3); np.random.seed(3)
        for _ in range(6):
            loop.run_step()
            loop.step += 1
        torch.cuda.synchronize()
        st = dict(loop.exchange.stats)
        n_b = len(loop.arena.bucket_ranges)
        ok = (st["exchanges"] == 6 and st["buckets_behind_event"] == (n_b - 1) * 6 and st["buckets_behind_graph_end"] == 6
              and not loop.exchange.flags.timed_out() and loop._graph_state.get("graph") is not None
              and bool(torch.isfinite(loop.arena.p).all()))
        loop.exchange.collect_timing()
        q.put(("ok" if ok else "bad", st, [round(x, 3) for x in loop.exchange.exposed_ms]))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("fail", traceback.format_exc(), None))
        raise


def test_exchange_mechanics_on_rccl():
    """The bucketed exchange with the REAL collective backend on one GPU: graph capture next to RCCL's watchdog thread,
    collectives on RCCL's own stream ordered behind the counter-polling kernel, the optimizer behind the collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=420)
    p.join(60)
    print(res)
    assert res[0] == "ok", res
    assert p.exitcode == 0


def _timeout_worker(port, q):
    """One rank on RCCL with the exchange forced on; after the micro-step has become a replayed graph, one bucket's wait is
    made to expect a signal that never comes.  The step must leave parameters, moments and EMA untouched (the optimizer
    launch reads the timed-out word) and the failure must surface as a RuntimeError no later than the next step."""
    try:
        for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0", LFVDM_FORCE_EXCHANGE="1", LFVDM_FLAG_TIMEOUT_S="0.3")
        os.environ.pop("LFVDM_DIST_BACKEND", None)
        import torch.distributed as dist
        from improved_diffusion import dist_util, script_util as su
        from improved_diffusion.train_util import TrainLoop
        from test_oracle_golden import load_case
        from test_forward_gpu import build_native
        dist_util.setup_dist()
        cfg, sd, _ = load_case("micro")
        model = build_native(cfg, sd).train()
        diffusion = su.create_gaussian_diffusion(steps=1000, rescale_timesteps=True, rescale_learned_sigmas=True)
        loop = TrainLoop(model=model, diffusion=diffusion, data=_data(2, 12, 4, 16, 7), batch_size=2, microbatch=-1,
                         lr=1e-3, ema_rate="0.9", log_interval=1000, save_interval=10 ** 9, resume_checkpoint="", use_fp16=False,
                         diffusion_space_kwargs={}, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.01,
                         lr_anneal_steps=0, sample_interval=None, pad_with_random_frames=True, max_frames=4,
                         enc_dec_chunk_size=20, args=argparse.Namespace(resume_id=""))
        ex = loop.exchange
        assert ex.overlap and ex.overlap_probe["ok"], ex.overlap_probe
        torch.manual_seed(3); np.random.seed(3)
        for _ in range(4):
            loop.run_step()
            loop.step += 1
        torch.cuda.synchronize()
        assert loop._graph_state.get("graph") is not None and not ex.flags.timed_out()
        before = [loop.arena.p.clone(), loop.exp_avg.clone(), loop.exp_avg_sq.clone(), loop.ema_flat[0].clone()]
        ex.micro_steps += 1                 # the waits of the next exchange expect one signal more than the graph will send
        raised_now = False
        try:
            loop.run_step()                 # early buckets give up after 0.3 s each; the optimizer launch must skip
            loop.step += 1
        except RuntimeError:
            raised_now = True
        torch.cuda.synchronize()
        after = [loop.arena.p, loop.exp_avg, loop.exp_avg_sq, loop.ema_flat[0]]
        unchanged = all(torch.equal(a, b) for a, b in zip(before, after))
        raised_next = False
        if not raised_now:
            try:
                loop.run_step()
            except RuntimeError as e:
                raised_next = "timed out" in str(e)
        torch.cuda.synchronize()
        still = all(torch.equal(a, b) for a, b in zip(before, [loop.arena.p, loop.exp_avg, loop.exp_avg_sq, loop.ema_flat[0]]))
        q.put(("ok" if (unchanged and still and (raised_now or raised_next) and ex.timeouts_seen == 1 and not ex.flags.timed_out()
                       and not ex.overlap) else "bad",
               dict(unchanged=unchanged, still=still, raised_now=raised_now, raised_next=raised_next, probe=ex.overlap_probe)))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("fail", traceback.format_exc()))
        raise


def test_timed_out_bucket_wait_skips_the_optimizer_and_raises():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_timeout_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=420)
    p.join(60)
    print(res)
    assert res[0] == "ok", res


def _one_rank_times_out_worker(rank, world, port, q, one_step=False):
    """World 2 (gloo, shared card): after the micro-step has become a replayed graph ONLY RANK 1's bucket waits are made
    to give up.  Rank 1 then all-reduces a bucket its backward may not have finished; the skip decision must be
    collective (the skip word rides in the last bucket's SUM all-reduce): BOTH ranks leave parameters / moments / EMA untouched, BOTH raise
    in the same later step, and after catching the error both carry on (exchange behind the graph's end, word cleared)
    with replicas that stay identical.
    one_step: the waits are mis-set for ONE step only (a one-off hiccup): the step after it would reduce cleanly - and must
    still be skipped by both ranks, because rank 1's sticky word keeps ITS optimizer launch from applying until the host
    has examined the word (one step late) and reset it."""
    try:
        for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), LFVDM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
                          LFVDM_FLAG_TIMEOUT_S="0.3")
        import torch.distributed as dist
        from improved_diffusion import dist_util, script_util as su
        from improved_diffusion.train_util import TrainLoop
        from test_oracle_golden import load_case
        from test_forward_gpu import build_native
        dist_util.setup_dist()
        cfg, sd, _ = load_case("micro")
        model = build_native(cfg, sd).train()
        diffusion = su.create_gaussian_diffusion(steps=1000, rescale_timesteps=True, rescale_learned_sigmas=True)
        loop = TrainLoop(model=model, diffusion=diffusion, data=_data(2, 12, 4, 16, 50 + rank), batch_size=2, microbatch=-1,
                         lr=1e-3, ema_rate="0.9", log_interval=1000, save_interval=10 ** 9, resume_checkpoint="", use_fp16=False,
                         diffusion_space_kwargs={}, fp16_scale_growth=1e-3, schedule_sampler=None, weight_decay=0.01,
                         lr_anneal_steps=0, sample_interval=None, pad_with_random_frames=True, max_frames=4,
                         enc_dec_chunk_size=20, args=argparse.Namespace(resume_id=""))
        ex = loop.exchange
        assert ex.overlap and ex.flags is not None, ex.overlap_probe
        torch.manual_seed(100 + rank); np.random.seed(100 + rank)
        for _ in range(4):
            loop.run_step()
            loop.step += 1
        torch.cuda.synchronize()
        assert loop._graph_state.get("graph") is not None and not ex.flags.timed_out()
        state = lambda: [loop.arena.p, loop.exp_avg, loop.exp_avg_sq, loop.ema_flat[0]]
        before = [t.clone() for t in state()]
        if rank == 1:
            ex.micro_steps += 1             # rank 1 only: its waits expect a signal the graph never sends
        raised_at = None
        for i in range(3):                  # the step with the timeout, then the step that must report it
            try:
                loop.run_step()
                loop.step += 1
            except RuntimeError as e:
                assert "timed out" in str(e)
                raised_at = i
                break
            if one_step and rank == 1 and i == 0:
                ex.micro_steps -= 1         # the hiccup is over: the next step's waits are satisfied by the graph
        torch.cuda.synchronize()
        unchanged = all(torch.equal(a, b) for a, b in zip(before, state()))
        if rank == 1 and not (one_step and raised_at != 0):
            ex.micro_steps -= 1
        # carry on after the error: no device-side waits any more, the word is cleared, the optimizer applies again
        cleared = not ex.flags.timed_out() and not ex.overlap
        for _ in range(2):
            loop.run_step()
            loop.step += 1
        torch.cuda.synchronize()
        moved = not torch.equal(before[0], loop.arena.p)
        both = [torch.empty_like(loop.arena.p) for _ in range(world)]
        dist.all_gather(both, loop.arena.p.detach())
        same = torch.equal(both[0], both[1])
        at = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
        dist.all_gather(at, torch.tensor([-1 if raised_at is None else raised_at], dtype=torch.int64, device="cuda"))
        same_step = int(at[0]) == int(at[1]) and int(at[0]) >= 0
        ok = unchanged and cleared and moved and same and same_step and ex.timeouts_seen == 1
        dist.barrier()
        q.put((rank, "ok" if ok else "bad", dict(unchanged=unchanged, cleared=cleared, moved=moved, same=same,
                                                  raised_at=[int(a) for a in at], seen=ex.timeouts_seen)))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
        raise


@pytest.mark.parametrize("one_step", [False, True])
def test_one_rank_timing_out_makes_every_rank_skip_and_raise_together(one_step):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_one_rank_times_out_worker, args=(r, world, port, q, one_step)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in sorted(res):
        print(r)
    assert all(r[1] == "ok" for r in res), res
