"""Data-parallel gradient exchange of ``TrainLoop``: what ``DistributedDataParallel`` does for the reference
(train_util.py:116-125: parameter broadcast at construction, bucketed all-reduce overlapped with the backward, mean over
ranks), re-plumbed for one process per MI355X and a backward pass that is ONE replayed hipGraph.

Layout.  The gradient arena (``ParamArena``) is laid out in BUCKETS in the order in which the backward pass finishes
them: the U-Net's stages are walked from the output head back to the input convolution and cut into ``n_buckets - 1``
groups of about equal size; the last bucket holds the remaining stages plus every parameter whose gradient is only
complete at the end of the backward pass (the timestep-embedding MLP, the FiLM projections and the RPE networks:
their gradients are produced by grouped launches in end-of-backward callbacks, ``_backward._EmbedNet``).  Each bucket is
a contiguous slice of the arena, so its exchange is one all-reduce with no packing copy.

Overlap.  ``UNetFunction`` plants an identity autograd node at the input of the first stage of every bucket.  Its
backward runs when all gradient kernels of the bucket have been issued: it folds the bucket's packed 3x3 weight-gradient
accumulators into the arena and bumps a counter in device memory (``_native.StreamFlags`` -> lfvdm_flag_add: an
ordinary kernel node, so it also fires in every REPLAY of the captured micro-step, where no Python runs.  An external
event-record node would be the textbook tool; PyTorch-ROCm and its HIP runtime refuse it under capture).  After the last
micro-batch the collective of bucket k is issued on a side stream behind lfvdm_flag_wait(counter k >= number of
micro-steps so far) and runs (RCCL over xGMI) while the rest of the backward graph is still executing.  Only the last bucket's collective is exposed.  The SUM is turned into the mean by ``grad_scale = 1/world`` in
the fused optimizer.

The class is device-agnostic: on CPU tensors (gloo, the world-size-2 CPU test) there are no streams or events and the
collectives are issued in bucket order.
"""
import os

import torch as th
import torch.distributed as dist

_LATE_MARKS = ("time_embed.", ".emb_layers.", ".rpe_q.", ".rpe_k.", ".rpe_v.")


def is_late(name):
    """Parameters whose gradients are written by the end-of-backward callbacks."""
    return name.startswith(_LATE_MARKS[0]) or any(m in name for m in _LATE_MARKS[1:])


def stage_of(name):
    """Position of a parameter's stage along the forward chain as a sortable key: input_blocks.i -> (0, i),
    middle_block -> (1, 0), output_blocks.i -> (2, i), out -> (3, 0)."""
    head = name.split(".")
    if head[0] == "input_blocks":
        return (0, int(head[1]))
    if head[0] == "middle_block":
        return (1, 0)
    if head[0] == "output_blocks":
        return (2, int(head[1]))
    if head[0] == "out":
        return (3, 0)
    return None            # time_embed: late


def plan_buckets(named_params, n_buckets=5):
    """-> (groups, marks): ``groups`` = parameter indices per bucket, in layout order (bucket 0 is finished first by the
    backward pass); ``marks`` = {stage key: bucket} for the stages whose INPUT carries the bucket's marker node.

    The last bucket holds ONLY what is complete at the very end of the backward pass: the parameters whose gradients come
    from the end-of-backward grouped launches (``is_late``) and the input convolution (stage (0, 0): its weight gradient
    is the last kernel of the U-Net's backward; no marker can sit in front of it - the network input carries no
    gradient).  Every other stage belongs to one of the ``n_buckets - 1`` early buckets, whose collectives start INSIDE
    the backward pass: the stages are walked from the output head back to input_blocks.1 and cut into groups of about
    equal size, the last group's marker sitting at the input of input_blocks.1.  (Round 4 left the stages behind the third
    cut in the last bucket as well: 44.5 of 122 MB that could not start before the graph's end at the training shape.)"""
    named = list(named_params)
    stages = {}
    for i, (n, p) in enumerate(named):
        s = stage_of(n)
        if s is not None and not is_late(n):
            stages.setdefault(s, []).append(i)
    order = [s for s in sorted(stages, reverse=True) if s != (0, 0)]          # backward order: head first; (0, 0) is the tail's
    late = [i for i, (n, _) in enumerate(named) if stage_of(n) is None or is_late(n)]
    tail = stages.get((0, 0), []) + late
    n_early = min(max(0, n_buckets - 1), len(order))
    total = sum(named[i][1].numel() for s in order for i in stages[s])
    groups, marks, cur, acc = [], {}, [], 0
    target = total / n_early if n_early else 0
    for pos, s in enumerate(order):
        cur += stages[s]
        acc += sum(named[i][1].numel() for i in stages[s])
        left = len(order) - 1 - pos                           # stages still to place
        last_group = len(groups) == n_early - 1
        # cut here (the marker sits at the input of stage s): the group is full and later groups can still get a stage each;
        # the last early group runs to the end of the walk
        if n_early and ((not last_group and acc >= target and left >= n_early - 1 - len(groups)) or left == 0):
            marks[s] = len(groups)
            groups.append(cur)
            cur, acc = [], 0
    if n_early == 0:        # one bucket: everything behind the graph's end
        tail = cur + tail
    groups.append(tail)
    groups = [g for g in groups if g]
    assert sorted(i for g in groups for i in g) == list(range(len(named)))
    return groups, marks


def _comm_priority():
    """Priority of the communication stream (LFVDM_COMM_STREAM_PRIORITY, default 0 = normal)."""
    return int(os.environ.get("LFVDM_COMM_STREAM_PRIORITY", "0"))


class _Mark(th.autograd.Function):
    """Identity; its backward tells the exchange that every gradient kernel of a bucket has been issued."""

    @staticmethod
    def forward(ctx, h, exchange, k):
        ctx.exchange, ctx.k = exchange, k
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        ctx.exchange.bucket_ready(ctx.k)
        return g, None, None


class GradExchange:
    def __init__(self, arena, marks, world=None, overlap=None):
        self.arena = arena
        self.marks = dict(marks)
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.ranges = list(arena.bucket_ranges)
        self.n_early = len(self.ranges) - 1
        self.on_gpu = arena.g.is_cuda
        if overlap is None:
            overlap = os.environ.get("LFVDM_OVERLAP_EXCHANGE", "1") != "0"
        self.overlap = bool(overlap) and self.on_gpu and self.n_early > 0
        self.bucket_param_ids = [frozenset(id(arena.params[i]) for i in g) for g in arena.groups]
        self.comm = None
        self.flags = None
        self.fired = [False] * self.n_early
        self.micro_steps = 0               # every micro-step (eager or replayed) bumps every early counter once
        self.wait_timeout_s = float(os.environ.get("LFVDM_FLAG_TIMEOUT_S", "20"))
        self.overlap_probe = None
        self.timeouts_seen = 0
        self._late = []                    # (pinned word, event): the timed-out word of earlier steps, read one step late
        if self.overlap:
            from . import _native as nat
            # HIP maps streams of one priority onto a few hardware queues in creation order: a fresh stream may alias the
            # queue that replays the backward graph (a wait parked there would only start after the graph).  Probe, and on a
            # miss try the next stream - the rejected ones are kept alive meanwhile so that the runtime moves on to another
            # queue.  (Round 3 side-stepped this with a high-priority stream, which has its own queue - and was measured in
            # round 4 to delay every launch of the graph while a polling kernel sits on it.)
            rejected = []
            for attempt in range(8):
                self.comm = th.cuda.Stream(priority=_comm_priority())
                self.overlap_probe = self._probe_overlap()
                self.overlap_probe["streams_tried"] = attempt + 1
                if self.overlap_probe["ok"]:
                    break
                rejected.append(self.comm)
            del rejected
            # The decision must be the SAME on every rank: it changes what each rank enqueues (device-side waits in front
            # of the early collectives or not) - a rank whose probe failed would otherwise run a different schedule than
            # its peers against the same collectives.  MIN over the ranks: overlap only if every rank's probe passed.
            ok = bool(self.overlap_probe["ok"])
            if dist.is_initialized() and dist.get_world_size() > 1:
                everyone = self.agree(ok, arena.g.device)
                self.overlap_probe["ok_on_every_rank"] = everyone
                ok = ok and everyone
            if ok:
                self.flags = nat.StreamFlags(self.n_early, arena.g.device)
            else:               # a stream shares main's hardware queue (or a wait gave up) on some rank: exchange behind the graph's end
                self.overlap = False
        self._works = []
        self._timing = []              # (start, end) event pairs of the exposed waits, read lazily
        self.exposed_ms = []
        self.stats = {"exchanges": 0, "buckets_behind_event": 0, "buckets_behind_graph_end": 0}

    @staticmethod
    def agree(ok, device):
        """True iff ``ok`` holds on EVERY rank (MIN all-reduce): decisions that change what a rank enqueues against the shared
        sequence of collectives must be the same everywhere."""
        t = th.tensor([1 if ok else 0], device=device, dtype=th.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # ------------------------------------------------------------------ overlap: probe and failure handling
    def _probe_overlap(self):
        """Can a wait parked on the communication stream finish WHILE the main stream is still busy?  HIP maps user
        streams onto a few hardware queues; if ``comm`` aliases the queue that replays the backward graph, every early
        bucket would only start after the graph (no overlap) - or, with a wait dispatched in front of its signal, stall
        until the timeout.  The probe replays the pattern once with harmless kernels: main = [signal, 5 ms busy kernel],
        comm = [wait for the signal].  Overlap is enabled only if the wait returns, without its timeout, at least a
        millisecond before main's busy kernel ends."""
        from . import _native as nat
        dev = self.arena.g.device
        sig, busy = nat.StreamFlags(1, dev), nat.StreamFlags(1, dev)
        main = th.cuda.current_stream()
        e_comm, e_main = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        th.cuda.synchronize(dev)
        with th.cuda.stream(self.comm):
            sig.wait(0, 1, self.comm, timeout_s=0.5)        # parked BEFORE its signal exists: the bad order on purpose
            e_comm.record(self.comm)
        sig.add(0)
        busy.wait(0, 1, main, timeout_s=0.005)             # never signalled: a 5 ms busy kernel (its timed-out word is private)
        e_main.record(main)
        th.cuda.synchronize(dev)
        lead = e_comm.elapsed_time(e_main)
        ok = (not sig.timed_out()) and lead > 1.0
        if os.environ.get("LFVDM_TEST_PROBE_FAIL_RANK", "") == os.environ.get("RANK", "0"):
            ok = False          # test hook: this rank's probe "fails" (tests/test_dist_gpu.py: the decision must stay collective)
        return {"ok": bool(ok), "wait_returned_ms_before_main_idle": round(float(lead), 3), "wait_timed_out": bool(sig.timed_out())}

    def skip_flag_ptr(self):
        """Device word the optimizer launch checks (lfvdm_adamw_args.skip_flag): non-zero once a bucket wait of THIS rank
        gave up (sticky until ``reset_timeout``)."""
        return self.flags.timed_out_ptr() if self.flags is not None else None

    def skip_word(self):
        """The float behind the last gradient bucket (``ParamArena.g_full[numel]``): a timed-out wait raises it to 1.0 on
        the rank it happened on, the last bucket's SUM all-reduce - which carries it as one extra element - makes it
        non-zero on EVERY rank before the optimizer launch reads it (lfvdm_adamw_args.skip_flag2), ``zero_grad`` clears it
        with the gradients.  No collective of its own (round 4 spent a separate MAX all-reduce per step on this)."""
        gf = getattr(self.arena, "g_full", None)
        return gf[self.arena.numel:self.arena.numel + 1] if gf is not None else None

    def skip_word_ptr(self):
        w = self.skip_word()
        return w.data_ptr() if w is not None else None

    def poll_timeout(self, sync=False):
        """Once per optimizer step, after the optimizer launch: raises if a bucket's wait on the backward graph timed out
        on ANY rank in the PREVIOUS step (sync=True: in any step so far).  The decision is collective without a collective
        of its own: a wait that gives up raises this rank's STICKY int word (``skip_flag`` of its optimizer launch) and the
        float word behind the last gradient bucket, which rides in that bucket's SUM all-reduce and is ``skip_flag2`` of
        EVERY rank's optimizer launch; while the sticky word is up every later wait of that rank raises the riding word
        again (``zero_grad`` clears it each step), so the replicas skip TOGETHER until the word has been examined - by every
        rank at exactly step s + 1 for the word of step s (fixed lag: its event is normally long complete when the host gets
        here, the host runs at most one step ahead of the GPU anyway) - and ``reset_timeout`` has cleared it.  All ranks
        raise in the same step, nobody is left alone in a collective; parameters, moments and EMA are those of the last good
        step."""
        if self.flags is None:
            return
        if sync:
            bad = self.flags.timed_out()
        else:
            host = th.empty(2, dtype=th.int32).pin_memory()
            host[:1].copy_(self.flags.buf[self.flags.n, :1], non_blocking=True)
            w = self.skip_word()                    # (bits of the float: any rank's timeout, after the reduce)
            if w is not None:
                host[1:].copy_(w.view(th.int32), non_blocking=True)
            else:
                host[1] = 0
            ev = th.cuda.Event()
            ev.record()
            self._late.append((host, ev))
            bad = False
            while len(self._late) > 1:              # everything but the word just enqueued: fixed one-step lag
                h, e = self._late.pop(0)
                e.synchronize()
                bad = bad or bool(h[0].item()) or bool(h[1].item())
        if bad:
            self.reset_timeout()
            self.skipped_steps = 2
            raise RuntimeError("gradient exchange: a bucket's wait on the backward graph timed out (lfvdm_flag_wait) on "
                               "at least one rank.  Every rank's optimizer launch skipped that step and the one after it "
                               "(the word is examined one step late, by every rank in the same step): parameters, moments "
                               "and EMA are those of the last good step; TrainLoop.opt_step has been taken back by the two "
                               "skipped steps")

    def reset_timeout(self):
        """After a timeout has been reported: whoever catches the error and carries on gets the exchange behind the
        graph's end (no device-side waits any more), with a CLEARED word - otherwise every later optimizer launch would
        skip as well.  Collective by construction: every rank gets here in the same step (poll_timeout)."""
        self.overlap = False
        self.timeouts_seen += 1
        self._late = []
        if self.flags is not None:
            th.cuda.current_stream().wait_stream(self.comm)
            self.flags.buf[self.flags.n].zero_()

    # ------------------------------------------------------------------ construction-time sync
    def broadcast(self, *flats):
        """Same replica everywhere (DDP does this at construction, reference train_util.py:116-125)."""
        if self.world > 1:
            for f in flats:
                dist.broadcast(f, 0)

    # ------------------------------------------------------------------ hooks from the backward pass
    def mark(self, h, stage_key):
        """Plant the marker of the bucket that starts at ``stage_key`` (no-op for other stages)."""
        k = self.marks.get(stage_key)
        if k is None or not h.requires_grad:
            return h
        return _Mark.apply(h, self, k)

    def bucket_ready(self, k):
        from ._backward import _packed
        _packed.flush(only=self.bucket_param_ids[k])          # 3x3 weight gradients of this bucket -> arena
        if self.overlap:
            self.flags.add(k)             # a kernel node of the graph when the micro-step is being captured
        if self.overlap:
            self.fired[k] = True

    # ------------------------------------------------------------------ after the last micro-batch
    def launch(self):
        """Issue one SUM all-reduce per bucket.  GPU: on the side stream, bucket k behind its event, the last bucket
        (and any bucket whose marker did not fire) behind everything enqueued so far."""
        if self.world <= 1:
            return
        self.stats["exchanges"] += 1
        g = self.arena.g
        if not self.on_gpu:
            for k, (lo, hi) in enumerate(self.ranges):
                if hi > lo:     # (the last bucket carries the skip word here too: the same sequence of collectives as on the GPU)
                    last = k == len(self.ranges) - 1 and self.skip_word() is not None
                    dist.all_reduce(self.arena.g_full[lo:self.arena.numel + 4] if last else g[lo:hi], op=dist.ReduceOp.SUM)
            return
        if self.comm is None:
            self.comm = th.cuda.Stream(priority=_comm_priority())
        main = th.cuda.current_stream()
        end = th.cuda.Event()
        end.record(main)
        with th.cuda.stream(self.comm):
            tail_started = False
            for k, (lo, hi) in enumerate(self.ranges):
                if hi <= lo:
                    continue
                early = k < self.n_early and self.overlap and self.fired[k] and not tail_started
                if early:
                    self.flags.wait(k, self.micro_steps, self.comm, timeout_s=self.wait_timeout_s, also_f32=self.skip_word_ptr())
                    self.stats["buckets_behind_event"] += 1
                else:
                    if not tail_started:
                        self.comm.wait_event(end)
                        tail_started = True
                    self.stats["buckets_behind_graph_end"] += 1
                buf = g[lo:hi]
                if k == len(self.ranges) - 1 and self.skip_word() is not None:
                    # collective skip decision without a collective of its own: the last bucket carries the skip word
                    # (ParamArena.g_full[numel], raised to 1.0 by a wait that gave up on this rank) as one extra element
                    # of its SUM - non-zero on every rank afterwards if ANY rank let a half-written bucket into the sums.
                    # Every rank does this whatever its own overlap state: the sequence of collectives is rank-invariant
                    buf = self.arena.g_full[lo:self.arena.numel + 4]
                self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
    def micro_step_done(self):
        """Called by TrainLoop after every micro-step it has enqueued (eager or graph replay)."""
        self.micro_steps += 1

    def wait(self):
        """Order the current stream (the optimizer comes next) behind the collectives; the time it stalls is the
        exposed part of the exchange."""
        if not self._works:
            return
        if self.on_gpu:
            a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            a.record()
            for w in self._works:
                w.wait()
            th.cuda.current_stream().wait_stream(self.comm)
            b.record()
            self._timing.append((a, b))
            if len(self._timing) > 64:
                self.collect_timing()
        else:
            for w in self._works:
                if w is not None:
                    w.wait()
        self._works = []

    def collect_timing(self):
        """Exposed-exchange times (ms) of the steps whose events have completed; never synchronises."""
        keep = []
        for a, b in self._timing:
            if b.query():
                self.exposed_ms.append(a.elapsed_time(b))
            else:
                keep.append((a, b))
        self._timing = keep
        return self.exposed_ms
