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


def plan_buckets(named_params, n_buckets=4):
    """-> (groups, marks): ``groups`` = parameter indices per bucket, in layout order (bucket 0 is finished first by the
    backward pass); ``marks`` = {stage key: bucket} for the stages whose INPUT carries the bucket's marker node."""
    named = list(named_params)
    stages = {}
    for i, (n, p) in enumerate(named):
        s = stage_of(n)
        if s is not None and not is_late(n):
            stages.setdefault(s, []).append(i)
    order = sorted(stages, reverse=True)                      # backward order: head first
    total = sum(named[i][1].numel() for idx in stages.values() for i in idx)
    late = [i for i, (n, _) in enumerate(named) if stage_of(n) is None or is_late(n)]
    n_early = max(0, n_buckets - 1)
    groups, marks, cur, acc, target = [], {}, [], 0, (total / n_buckets if n_buckets else 0)
    for pos, s in enumerate(order):
        if len(groups) < n_early:
            cur += stages[s]
            acc += sum(named[i][1].numel() for i in stages[s])
            last_stage = pos == len(order) - 1
            if acc >= target and not last_stage:              # cut here: the marker sits at the input of stage s
                marks[s] = len(groups)
                groups.append(cur)
                cur, acc = [], 0
        else:
            cur += stages[s]
    groups.append(cur + late)                                # finished at the end of the backward pass
    assert sorted(i for g in groups for i in g) == list(range(len(named)))
    return groups, marks


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
        if self.overlap:
            from . import _native as nat
            self.comm = th.cuda.Stream()
            self.flags = nat.StreamFlags(self.n_early, arena.g.device)
        self._works = []
        self._timing = []              # (start, end) event pairs of the exposed waits, read lazily
        self.exposed_ms = []
        self.stats = {"exchanges": 0, "buckets_behind_event": 0, "buckets_behind_graph_end": 0}

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
        from ._backward import _packed, _side

        def fold_and_signal():
            _packed.flush(only=self.bucket_param_ids[k])          # 3x3 weight gradients of this bucket -> arena
            if self.overlap:
                self.flags.add(k)             # a kernel node of the graph when the micro-step is being captured
        # on the weight-gradient stream (behind the bucket's wgrad launches there and behind everything the main
        # chain has issued so far): the main chain itself never waits for a weight gradient before the end
        _side.flush_pending()
        _side.run(fold_and_signal, now=True)
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
            for lo, hi in self.ranges:
                if hi > lo:
                    dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM)
            return
        if self.comm is None:
            self.comm = th.cuda.Stream()
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
                    self.flags.wait(k, self.micro_steps, self.comm)
                    self.stats["buckets_behind_event"] += 1
                else:
                    if not tail_started:
                        self.comm.wait_event(end)
                        tail_started = True
                    self.stats["buckets_behind_graph_end"] += 1
                self._works.append(dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
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
