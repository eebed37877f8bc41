"""Training path of ``UNetVideoModel``: differentiable blocks on the gfx950 kernels.

Every block below is a ``torch.autograd.Function`` whose forward AND backward run the hand-written
HIP kernels for the heavy work of the reference's ``loss.backward()`` (train_util.py:328):

  * 3x3 / 1x1 convolutions and the qkv / proj_out linears: forward ``lfvdm_conv_igemm``, data gradient
    ``lfvdm_conv_igemm`` on transposed-flipped weights (``lfvdm_pack_conv_weight_t``; stride-2 convs use
    the zero-insertion gather), weight/bias gradient ``lfvdm_conv_wgrad`` (fp32 MFMA, the fused
    GroupNorm/FiLM/SiLU operand is recomputed on the fly, never stored);
  * GroupNorm(+FiLM)(+SiLU): ``lfvdm_gn_apply`` forward (normalised + activated tensor materialised once and
    saved: it is the raw operand of the GEMM and of the weight-gradient kernel), ``lfvdm_gn_bwd_stats/apply`` backward;
    temporal GroupNorm ``lfvdm_gn_temporal(_bwd)``;
  * attention cores: forward ``lfvdm_attn_spatial`` (saves the log-sum-exp) / ``lfvdm_attn_temporal``, backward
    ``lfvdm_attn_spatial_bwd`` (flash style dq / dk,dv kernels) and ``lfvdm_attn_temporal_bwd`` (rows / cols /
    rpe kernels: dqkv and the three R gradients, no atomics);
  * the C x C output layer of every RPE network on the same GEMM / wgrad kernels.

3x3 weight gradients accumulate in packed [Cout][tap][Cin] buffers and are folded into the OIHW ``.grad`` tensors
by ONE grouped kernel at the end of each backward pass (``_PackedGrads``).  Launch shapes are tuned on first sight
(``_native.tuned_code``).  What is left on library ops (all on the GPU, no CPU path): the (B x 4ch) time-embedding /
FiLM projections and the 3-feature / time projections inside the RPE networks - a few (2..800)-row GEMMs per
block, < 1 % of the FLOPs - plus their elementwise glue (SiLU, log1p features) and GroupNorm parameter-gradient
reductions.
"""
import ctypes
import os
import weakref

import torch as th
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nat
from .nn import timestep_freqs

_EPS = 1e-5


def _new(*shape, like):
    return th.empty(*shape, device=like.device, dtype=th.float32)


# ----------------------------------------------------------------------------- packed weights
class _WeightPacks:
    """Persistent packed copies of the conv weights of leaf parameters: forward layout [Cout][tap][Cin] and
    data-gradient layout [Cin][tap][Cout] (flipped taps).  All of them are refreshed by ONE grouped launch the first
    time any is requested after the parameters changed (optimizer step / load_state_dict) instead of two small
    launches per convolution per step.  Parameters are held weakly: entries of freed models are dropped."""

    def __init__(self):
        self.ent = {}           # (data_ptr, shape, transposed) -> [weakref(base param), packed, stamp]
        self.table = None

    @staticmethod
    def _stamp(base):
        return (nat.param_epoch[0], base._version)

    def _pack_one(self, key, out):
        ptr_, (Cout, Cin, k, _), tr, ld = key
        if ld:      # zero-padded destination (input / output conv): rare single packs go through the library ops
            e = self.ent.get(key)
            base = e[0]() if e is not None else None
            off = (ptr_ - base.data_ptr()) // 4
            w4 = base.detach().reshape(-1)[off:off + Cout * Cin * k * k].view(Cout, Cin, k, k)
            if tr:
                out[:, :, :Cout].copy_(w4.flip(2, 3).reshape(Cout, Cin, k * k).permute(1, 2, 0))
            else:
                out[:, :, :Cin].copy_(w4.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin))
            return
        fn = nat.lib().lfvdm_pack_conv_weight_t if tr else nat.lib().lfvdm_pack_conv_weight
        nat.check(fn(ptr_, out.data_ptr(), Cout, Cin, k, nat.stream()), "lfvdm_pack_conv_weight")

    def get(self, w4, transposed, ld=0):
        """ld: channel stride of the packed copy when it is wider than the weight (zero padding channels: the 5-channel
        input conv and the 4-filter output conv are staged as 32-channel operands); leaf parameters only."""
        base = w4._base if w4._base is not None else w4
        Cout, Cin, k, _ = w4.shape
        if not isinstance(base, nn.Parameter) or not w4.is_contiguous():
            assert not ld, "padded packs are kept for leaf parameters only"
            out = _new(*((Cin, k * k, Cout) if transposed else (Cout, k * k, Cin)), like=w4)     # derived weight: pack now
            (nat.pack_conv_weight_t if transposed else nat.pack_conv_weight)(w4.contiguous(), out)
            return out
        key = (w4.data_ptr(), tuple(w4.shape), bool(transposed), int(ld))
        e = self.ent.get(key)
        if e is None or e[0]() is not base:       # new weight (or the address was re-used by another parameter)
            shape = (Cin, k * k, ld or Cout) if transposed else (Cout, k * k, ld or Cin)
            out = th.zeros(*shape, device=w4.device, dtype=th.float32) if ld else _new(*shape, like=w4)
            self.ent[key] = [weakref.ref(base), out, self._stamp(base)]
            self._pack_one(key, out)
            self.table = None
            return out
        if e[2] != self._stamp(base):
            self.refresh_all()
        return e[1]

    def _live(self, key, e):
        """The entry's parameter is alive and its storage is still where (and what) the packed copy was made from."""
        base = e[0]()
        if base is None:
            return False
        ptr_, shape = key[0], key[1]
        n = 1
        for d in shape:
            n *= d
        lo = base.data_ptr()
        return lo <= ptr_ and ptr_ + 4 * n <= lo + 4 * base.numel()

    def refresh_if_stale(self):
        """Re-pack (one grouped launch) if any parameter changed since the packed copies were made.  Callers that
        replay captured launches - which run no Python and therefore never reach ``get`` - call this eagerly before
        each replay (TrainLoop._graphed_micro_step)."""
        if any(e[0]() is not None and e[2] != self._stamp(e[0]()) for e in self.ent.values()):
            self.refresh_all()

    def refresh_all(self):
        # entries of freed parameters, and of live parameters whose storage moved (ParamArena re-points p.data; .to()):
        # their recorded source pointer may be freed memory
        dead = [k for k, e in self.ent.items() if not self._live(k, e)]
        for k in dead:
            del self.ent[k]
        if dead:
            self.table = None
        if not self.ent:
            return
        if self.table is None:
            if th.cuda.is_current_stream_capturing():      # cannot upload a job table now: pack one by one
                for key, e in self.ent.items():
                    self._pack_one(key, e[1])
                    self._restamp(e)
                return
            jobs, blk = [], 0
            for (ptr_, (Cout, Cin, k, _k), tr, ld), e in self.ent.items():
                jobs.append(nat.PackJob(ptr_, e[1].data_ptr(), Cout, Cin, k * k, int(tr), blk, ld))
                blk += ((Cout + 31) // 32) * ((Cin + 31) // 32)
            dev = next(iter(self.ent.values()))[1].device
            self.table, self.blocks, self.njobs = nat.jobs_to_device(jobs, dev), blk, len(jobs)
        nat.check(nat.lib().lfvdm_pack_conv_weights(self.table.data_ptr(), self.njobs, self.blocks, nat.stream()),
                  "lfvdm_pack_conv_weights")
        for e in self.ent.values():
            self._restamp(e)

    def _restamp(self, e):
        base = e[0]()          # (a parameter of a dropped model may be collected while this method runs: pruned next time)
        if base is not None:
            e[2] = self._stamp(base)


_packs = _WeightPacks()


def _pack(w, ld=0):
    """OIHW -> [Cout][taps][Cin] (forward operand layout)."""
    return _packs.get(w, False, ld)


def _pack_t(w4, ld=0):
    """OIHW (or [O][I] as [O][I][1][1]) -> [Cin][taps][Cout], taps flipped (data-gradient operand)."""
    return _packs.get(w4, True, ld)


def _grad_of(p):
    """The parameter's gradient buffer (the arena view set up by TrainLoop, or a fresh zero tensor)."""
    if p.grad is None:
        p.grad = th.zeros_like(p)
    return p.grad


class _PackedGrads:
    """Packed [Cout][tap][Cin] accumulators of the 3x3 weights.  The wgrad kernels add their partial tiles with
    float atomics; in the packed layout consecutive lanes hit consecutive addresses, in OIHW every wave atomic
    would scatter over ~18 cache lines (measured 2-4x slower kernels).  At the end of each backward pass ONE
    grouped kernel folds all of them into the OIHW ``.grad`` tensors and zeroes them again."""

    def __init__(self):
        self.bufs = {}          # id(param) -> (weakref(param), packed buffer)
        self.tables = {}        # job tables by (accumulator, gradient) pointer set: one per distinct fold
        self.done = set()       # ids folded by partial folds of the running backward pass
        self.queued = False

    def buffer(self, w, ld=0):
        """ld: channel stride of the accumulator when the operand was zero-padded to more channels than the weight has."""
        ent = self.bufs.get(id(w))
        if ent is None or ent[0]() is not w or ent[1].device != w.device or ent[1].shape[2] != (ld or w.shape[1]):
            Cout, Cin, k, _ = w.shape
            ent = (weakref.ref(w), th.zeros(Cout, k * k, ld or Cin, device=w.device, dtype=th.float32))
            self.bufs[id(w)] = ent
            self.tables.clear()
        if not self.queued:     # fold at the end of the running backward pass (also under graph capture)
            self.queued = True
            th.autograd.Variable._execution_engine.queue_callback(self.flush)
        return ent[1]

    def flush(self, only=None):
        """Fold the packed accumulators into the OIHW ``.grad`` tensors (and re-zero them).  ``only``: a set of
        parameter ids - a partial fold in the middle of the backward pass, issued by the gradient exchange when a
        bucket is complete (_exchange.GradExchange.bucket_ready); the end-of-backward call then folds the rest."""
        if only is None:
            self.queued = False
        for k in [k for k, (r, _) in self.bufs.items() if r() is None]:     # parameters of freed models
            del self.bufs[k]
            self.tables.clear()
        done = self.done
        live = [(r(), gp) for k, (r, gp) in self.bufs.items() if (k in only if only is not None else k not in done)]
        live = [(w, gp) for w, gp in live if w is not None and w.grad is not None and w.grad.is_cuda]
        if only is None:
            self.done = set()
        else:
            self.done = done | {id(w) for w, _ in live}
        if not live:
            return
        key = tuple((gp.data_ptr(), w.grad.data_ptr()) for w, gp in live)
        ent = self.tables.get(key)
        if ent is None:
            if th.cuda.is_current_stream_capturing():
                raise RuntimeError("packed-gradient job table missing under stream capture: run the step eagerly once first")
            jobs, row0, mx = [], 0, 0
            for w, gp in live:
                Cout, Cin, k, _ = w.shape
                ldp = gp.shape[2]
                jobs.append(nat.UnpackJob(gp.data_ptr(), w.grad.data_ptr(), Cout, Cin, k * k, row0, ldp if ldp != Cin else 0, 0))
                row0 += Cout
                mx = max(mx, k * k * ldp)
            if len(self.tables) > 64:
                self.tables.clear()
            ent = self.tables[key] = (nat.jobs_to_device(jobs, live[0][0].device), len(jobs), row0, mx)
        table, njobs, rows, mx = ent
        nat.check(nat.lib().lfvdm_unpack_conv_grads(table.data_ptr(), njobs, rows, mx, nat.stream()), "lfvdm_unpack_conv_grads")


_packed = _PackedGrads()


class _GradMode:
    """How parameter gradients leave the backward pass.

    inplace=False (default): every Function returns its parameter gradients to autograd, so AccumulateGrad runs and
    hooks fire - DistributedDataParallel (reference train_util.py:116-125), ``register_hook``, gradient
    checkpointing wrappers all behave as with ordinary modules.
    inplace=True (``model.native_grad_accumulation = True``, set by ``TrainLoop``, which owns a flat gradient
    arena and reduces it itself): the kernels accumulate straight into ``p.grad`` / the packed accumulators and
    autograd sees no parameter gradients - ~800 small adds and the per-tensor temporaries disappear."""
    inplace = False


_mode = _GradMode()


def _leaf(*ps):
    return _mode.inplace and all(p is None or p.is_leaf for p in ps)


def _wgrad_accumulate(w, b, **kw):
    """Weight/bias gradient accumulated by the wgrad kernel without temporaries or AccumulateGrad adds: 1x1 /
    linear weights straight into ``w.grad`` (packed == OIHW), 3x3 weights into their packed accumulator, which
    is folded into ``w.grad`` once per backward pass (``_PackedGrads``)."""
    kw.pop("ksize", None)
    ld = kw.pop("ld", 0)
    k = w.shape[2] if w.dim() == 4 else 1
    gw = _grad_of(w)
    out = _packed.buffer(w, ld) if k == 3 else gw
    gb = _grad_of(b) if b is not None else None
    nat.conv_wgrad(out=out, bias=gb, Cout=w.shape[0], ksize=k, out_mode=0, **kw)


def _wgrad_into(grad_w_shape, like, **kw):
    """Run lfvdm_conv_wgrad into a zeroed packed buffer and return (dW in the parameter layout, db)."""
    Cout, Cin = grad_w_shape[0], grad_w_shape[1]
    k = grad_w_shape[2] if len(grad_w_shape) == 4 else 1
    kw.pop("ksize", None)
    gp = th.zeros(Cout, k * k, Cin, device=like.device, dtype=th.float32)
    db = th.zeros(Cout, device=like.device, dtype=th.float32)
    nat.conv_wgrad(out=gp, bias=db, Cout=Cout, ksize=k, **kw)
    if k == 1:
        return gp.view(grad_w_shape), db
    g = _new(*grad_w_shape, like=like)
    nat.unpack_conv_grad(gp, g, accumulate=False)
    return g, db


_SKIP_SLOTS = os.environ.get("LFVDM_NO_SKIP_SLOTS") is None      # A/B aid


class _SkipSlot:
    """Hand-off of a skip connection's gradient (in-place mode).  An encoder output h feeds the next stage AND a decoder
    ResBlock; autograd would sum the two gradients of h with an add kernel per skip.  Instead the decoder block (whose
    backward always runs first: it comes later in the forward pass) leaves its gradient here and returns nothing for h,
    and the kernel that produces the next stage's gradient of h adds it on the way out (``res`` of the data-gradient
    GEMM, ``add2`` of the fused GroupNorm backward)."""
    __slots__ = ("g",)
    pending = []            # slots filled in the running backward pass and not yet drained
    queued = False

    def __init__(self):
        self.g = None

    def put(self, g):
        """The decoder block leaves its gradient of the skip tensor.  The hand-off takes this gradient OUT of autograd, so
        it is checked: at the end of the backward pass every filled slot must have been drained by its consumer."""
        assert self.g is None, "skip-gradient slot filled twice in one backward pass"
        self.g = g
        _SkipSlot.pending.append(self)
        if not _SkipSlot.queued:
            _SkipSlot.queued = True
            th.autograd.Variable._execution_engine.queue_callback(_SkipSlot.check_drained)

    def take(self):
        g, self.g = self.g, None
        return g

    @staticmethod
    def reset_stale():
        """Start of a forward pass: autograd does not run the queued end-of-backward callbacks when a backward pass RAISES
        (out of memory, a kernel error), so ``queued`` would stay set and the leftover slots full - the next healthy backward
        pass would trip over them ("filled twice") or run without the drained check.  No backward pass is running here."""
        for s in _SkipSlot.pending:
            s.g = None
        _SkipSlot.pending, _SkipSlot.queued = [], False

    @staticmethod
    def check_drained():
        """End-of-backward callback: a slot that is still full means its consumer never ran or did not compute an input
        gradient (a partial ``torch.autograd.grad(..., inputs=subset)``, a first layer without an input gradient): the
        decoder's contribution would be dropped silently.  LFVDM_NO_SKIP_SLOTS=1 gives the plain autograd sums."""
        left = [s for s in _SkipSlot.pending if s.g is not None]
        _SkipSlot.pending, _SkipSlot.queued = [], False
        for s in left:
            s.g = None
        if left:
            raise RuntimeError(f"{len(left)} skip-connection gradient(s) were handed to a consumer that never ran in this "
                               "backward pass (partial backward?) - set LFVDM_NO_SKIP_SLOTS=1 for such calls")


# ----------------------------------------------------------------------------- plain conv (in / down / up)
class ConvFn(th.autograd.Function):
    """3x3 conv on channels-last rows [N*H*W][Cin] with optional stride 2 / nearest-2x upsample
    (Downsample / Upsample of reference unet.py:60-114)."""

    @staticmethod
    def forward(ctx, x, w, b, N, H, W, stride, up, cin_pad=0, x_slot=None):
        """cin_pad: the rows carry this many channels, of which the weight's Cin are real (the 5-channel input conv on
        32-channel rows; in-place gradient mode only: the padded operand packs belong to leaf parameters)."""
        Cout, Cin = w.shape[0], cin_pad or w.shape[1]
        Hin, Win = (2 * H, 2 * W) if up else (H, W)
        Ho, Wo = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
        out = _new(N * Ho * Wo, Cout, like=x)
        nat.conv_igemm(src0=x, C0=Cin, N=N, Hs=H, Ws=W, up=int(up), stride=stride, Ho=Ho, Wo=Wo, W=_pack(w, cin_pad), bias=b,
                       Cout=Cout, out=out, ldo=Cout)
        ctx.save_for_backward(x, w)
        ctx.params = (w, b) if _leaf(w, b) else None   # in-place mode: accumulate into .grad
        assert ctx.params is not None or not cin_pad
        ctx.geom = (N, H, W, stride, up, Ho, Wo)
        ctx.cin_pad = cin_pad
        ctx.x_slot = x_slot      # _SkipSlot: a gradient of x that the decoder leaves for this node to add
        assert x_slot is None or stride == 2
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        N, H, W, stride, up, Ho, Wo = ctx.geom
        Cout, Cin = w.shape[0], ctx.cin_pad or w.shape[1]
        dout = dout.contiguous()
        wkw = dict(src0=x, C0=Cin, N=N, Hs=H, Ws=W, up=int(up), stride=stride, Ho=Ho, Wo=Wo, res=dout, ldr=Cout)
        if ctx.params is not None:
            _wgrad_accumulate(ctx.params[0], ctx.params[1], ld=ctx.cin_pad, **wkw)
            dw = db = None
        else:
            dw, db = _wgrad_into(tuple(w.shape), x, **wkw)
        dx = None
        if ctx.x_slot is not None and ctx.x_slot.g is not None and not ctx.needs_input_grad[0]:
            raise RuntimeError("a skip-connection gradient was handed to a convolution that computes no input gradient")
        if ctx.needs_input_grad[0]:
            wt = _pack_t(w)
            if stride == 2:   # transposed conv: zero-insertion gather of dout on the (2Ho x 2Wo) grid
                assert 2 * Ho == H and 2 * Wo == W, "stride-2 data gradient needs even sizes"
                dx = _new(N * H * W, Cin, like=x)
                extra = ctx.x_slot.take() if ctx.x_slot is not None else None
                rkw = dict(res=extra, ldr=Cin) if extra is not None else {}
                nat.conv_igemm(src0=dout, C0=Cout, N=N, Hs=Ho, Ws=Wo, up=2, Ho=H, Wo=W, W=wt, Cout=Cin, out=dx, ldo=Cin, **rkw)
            else:
                Hin, Win = (2 * H, 2 * W) if up else (H, W)
                dfull = _new(N * Hin * Win, Cin, like=x)
                nat.conv_igemm(src0=dout, C0=Cout, N=N, Hs=Ho, Ws=Wo, Ho=Hin, Wo=Win, W=wt, Cout=Cin, out=dfull, ldo=Cin)
                if up:        # adjoint of nearest-2x: sum each 2x2 block
                    dx = dfull.view(N, H, 2, W, 2, Cin).sum(dim=(2, 4)).reshape(N * H * W, Cin).contiguous()
                else:
                    dx = dfull
        return dx, dw, db, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------- linear on rows
class LinearFn(th.autograd.Function):
    """y = x W^T + b (+ res) on rows [M][K] (nn.Linear qkv / proj_out, rpe.py:139,171)."""

    @staticmethod
    def forward(ctx, x, w, b, res):
        M, K = x.shape
        O = w.shape[0]
        y = _new(M, O, like=x)
        kw = dict(src0=x, C0=K, N=M, Hs=1, Ws=1, Ho=1, Wo=1, ksize=1, W=w, bias=b, Cout=O, out=y, ldo=O)
        if res is not None:
            kw.update(res=res, ldr=O)
        nat.conv_igemm(**kw)
        ctx.save_for_backward(x, w)
        ctx.params = (w, b) if (b is not None and _leaf(w, b)) else None
        ctx.has_res, ctx.has_bias = res is not None, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, K = x.shape
        O = w.shape[0]
        dy = dy.contiguous()
        wkw = dict(src0=x, C0=K, N=M, Hs=1, Ws=1, Ho=1, Wo=1, res=dy, ldr=O)
        if ctx.params is not None:
            _wgrad_accumulate(ctx.params[0], ctx.params[1], **wkw)
            dw = db = None
        else:
            dw, db = _wgrad_into((O, K), x, **wkw)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _new(M, K, like=x)
            nat.conv_igemm(src0=dy, C0=O, N=M, Hs=1, Ws=1, Ho=1, Wo=1, ksize=1, W=_pack_t(w.view(O, K, 1, 1)), Cout=K,
                           out=dx, ldo=K)
        return dx, dw, (db if ctx.has_bias else None), (dy if ctx.has_res else None)


# ----------------------------------------------------------------------------- GroupNorm helpers
def _gn_apply(a, b, C0, C1, N, P, gamma, beta, film, T, act):
    """GroupNorm(+FiLM)(+SiLU) applied once: -> (activated [N*P][C] tensor, coefA, coefB, stats).  The activated
    tensor feeds the implicit GEMM and (saved) the weight-gradient kernel as a raw operand (lfvdm_gn_apply)."""
    C = C0 + C1
    out = _new(N * P, C, like=a)
    cA, cB, stats = _new(N, C, like=a), _new(N, C, like=a), _new(N, 32, 2, like=a)
    L = nat.lib()
    need = int(L.lfvdm_gn_apply_ws_floats(C, N, P))        # > 0: large map, chunked two-launch form
    ws = _new(need, like=a) if need else None
    nat.check(L.lfvdm_gn_apply_ws(
        nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(gamma), nat.ptr(beta), film.data_ptr() if film is not None else None,
        T if film is not None else 1, film.stride(0) if film is not None else 0, _EPS, act, nat.ptr(out), nat.ptr(cA),
        nat.ptr(cB), nat.ptr(stats), nat.ptr(ws), need, nat.stream()), "lfvdm_gn_apply_ws")
    return out, cA, cB, stats


def _gn_forward(a, b, C0, C1, N, P, gamma, beta, film, T):
    C = C0 + C1
    cA, cB, stats = _new(N, C, like=a), _new(N, C, like=a), _new(N, 32, 2, like=a)
    nat.check(nat.lib().lfvdm_gn_coef_stats(
        nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(gamma), nat.ptr(beta), film.data_ptr() if film is not None else None,
        T if film is not None else 1, film.stride(0) if film is not None else 0, _EPS, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats),
        nat.stream()), "lfvdm_gn_coef_stats")
    return cA, cB, stats


def _gn_backward(da, a, b, C0, C1, N, P, cA, cB, stats, act, gamma, beta, film, T, want_dx=(True, True), dfilm_out=None,
                 inplace=False, add=None, add2=None):
    """Returns (dx_a, dx_b, dgamma, dbeta, dfilm).  da: [N*P][C] gradient w.r.t. act(GN(x)).  add, add2: optional
    [N*P][C] rows (further gradients of the same input) folded into dx by the in-place kernel."""
    C = C0 + C1
    if add2 is not None and not inplace:
        add, add2 = (add2 if add is None else add + add2), None
    L = nat.lib()
    dxa = _new(N * P, C0, like=da)
    dxb = _new(N * P, C1, like=da) if C1 else None
    need = int(L.lfvdm_gn_bwd_ws_floats(C, N, P))
    if need:
        # large map (pixel space, wide concats at 32x32): chunked two-launch form, thousands of workgroups
        return _gn_backward_chunked(da, a, b, C0, C1, N, P, cA, cB, stats, act, gamma, beta, film, T, dfilm_out, inplace, add,
                                    add2, dxa, dxb, need)
    if inplace and nat.deterministic():
        # fixed summation order: per-(sample, channel) sums (lfvdm_gn_bwd_stats), dx (lfvdm_gn_bwd_apply), then the
        # parameter / FiLM gradients by lfvdm_gn_param_grads, which walks the samples in order - no float atomics
        # (the statistics + dx launch is the training path's fused kernel, with the sums stored instead of added)
        sums = _new(N, C, 2, like=da)
        nat.check(L.lfvdm_gn_bwd_fused_sums(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats),
                                            act, nat.ptr(dxa), nat.ptr(dxb), nat.ptr(add), add.stride(0) if add is not None else 0,
                                            nat.ptr(add2), add2.stride(0) if add2 is not None else 0, nat.ptr(sums), nat.stream()),
                  "lfvdm_gn_bwd_fused_sums")
        dfilm = None
        if film is not None:
            dfilm = dfilm_out if dfilm_out is not None else th.zeros(N // T, 2 * C, device=da.device, dtype=th.float32)
        nat.check(L.lfvdm_gn_param_grads(nat.ptr(sums), nat.ptr(gamma), nat.ptr(beta), film.data_ptr() if film is not None else None,
                                         film.stride(0) if film is not None else 0, T, nat.ptr(_grad_of(gamma)), nat.ptr(_grad_of(beta)),
                                         dfilm.data_ptr() if dfilm is not None else None, dfilm.stride(0) if dfilm is not None else 0,
                                         N, C, nat.stream()), "lfvdm_gn_param_grads")
        return dxa, dxb, None, None, (None if dfilm_out is not None else dfilm)
    if inplace:
        # statistics, dx and the parameter gradients in ONE launch (float atomics into .grad / the FiLM gradient slot)
        dfilm = None
        if film is not None:       # accumulated into the caller's (zeroed) slot of the embedding network's gradient buffer
            dfilm = dfilm_out if dfilm_out is not None else th.zeros(N // T, 2 * C, device=da.device, dtype=th.float32)
        nat.check(L.lfvdm_gn_bwd_fused(
            nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats), act,
            nat.ptr(dxa), nat.ptr(dxb), nat.ptr(gamma), nat.ptr(beta),
            film.data_ptr() if film is not None else None, film.stride(0) if film is not None else 0, T,
            nat.ptr(_grad_of(gamma)), nat.ptr(_grad_of(beta)), dfilm.data_ptr() if dfilm is not None else None,
            dfilm.stride(0) if dfilm is not None else 0, nat.ptr(add), add.stride(0) if add is not None else 0,
            nat.ptr(add2), add2.stride(0) if add2 is not None else 0, nat.stream()), "lfvdm_gn_bwd_fused")
        return dxa, dxb, None, None, (None if dfilm_out is not None else dfilm)
    sums = _new(N, C, 2, like=da)
    nat.check(L.lfvdm_gn_bwd_stats(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats),
                                   act, nat.ptr(sums), nat.stream()), "lfvdm_gn_bwd_stats")
    nat.check(L.lfvdm_gn_bwd_apply(nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats),
                                   nat.ptr(sums), act, nat.ptr(dxa), nat.ptr(dxb), 0, 0, nat.stream()), "lfvdm_gn_bwd_apply")
    if add is not None:
        dxa = dxa + add[:, :C0]
        if dxb is not None:
            dxb = dxb + add[:, C0:]
    assert dfilm_out is None, "gradient slots belong to the in-place mode"
    dgamma, dbeta, dfilm = _gn_param_grads_torch(sums, gamma, beta, film, T, N, C)
    return dxa, dxb, dgamma, dbeta, dfilm


def _gn_backward_chunked(da, a, b, C0, C1, N, P, cA, cB, stats, act, gamma, beta, film, T, dfilm_out, inplace, add, add2,
                         dxa, dxb, need):
    """lfvdm_gn_bwd_ws for all three delivery modes: in place with float atomics (the training default), in place with a
    fixed summation order (LFVDM_DETERMINISTIC: per-(sample, channel) sums -> lfvdm_gn_param_grads), autograd (sums ->
    torch reductions).  (add2 has already been folded into add by the caller unless the kernel takes both.)"""
    C = C0 + C1
    L = nat.lib()
    ws = _new(need, like=da)
    atomics = inplace and not nat.deterministic()
    sums = None if atomics else _new(N, C, 2, like=da)
    dfilm = None
    if film is not None and inplace:
        dfilm = dfilm_out if dfilm_out is not None else th.zeros(N // T, 2 * C, device=da.device, dtype=th.float32)
    fptr = film.data_ptr() if film is not None else None
    fld = film.stride(0) if film is not None else 0
    nat.check(L.lfvdm_gn_bwd_ws(
        nat.ptr(da), nat.ptr(a), nat.ptr(b), C0, C1, N, P, nat.ptr(cA), nat.ptr(cB), nat.ptr(stats), act, nat.ptr(dxa),
        nat.ptr(dxb), nat.ptr(gamma), nat.ptr(beta), fptr if atomics else None, fld, T,
        nat.ptr(_grad_of(gamma)) if atomics else None, nat.ptr(_grad_of(beta)) if atomics else None,
        dfilm.data_ptr() if (atomics and dfilm is not None) else None, dfilm.stride(0) if dfilm is not None else 0,
        nat.ptr(add), add.stride(0) if add is not None else 0, nat.ptr(add2), add2.stride(0) if add2 is not None else 0,
        nat.ptr(sums), nat.ptr(ws), need, nat.stream()), "lfvdm_gn_bwd_ws")
    if atomics:
        return dxa, dxb, None, None, (None if dfilm_out is not None else dfilm)
    if inplace:
        nat.check(L.lfvdm_gn_param_grads(nat.ptr(sums), nat.ptr(gamma), nat.ptr(beta), fptr, fld, T, nat.ptr(_grad_of(gamma)),
                                         nat.ptr(_grad_of(beta)), dfilm.data_ptr() if dfilm is not None else None,
                                         dfilm.stride(0) if dfilm is not None else 0, N, C, nat.stream()), "lfvdm_gn_param_grads")
        return dxa, dxb, None, None, (None if dfilm_out is not None else dfilm)
    assert dfilm_out is None, "gradient slots belong to the in-place mode"
    dgamma, dbeta, dfilm = _gn_param_grads_torch(sums, gamma, beta, film, T, N, C)
    return dxa, dxb, dgamma, dbeta, dfilm


def _gn_param_grads_torch(sums, gamma, beta, film, T, N, C):
    s1, s2 = sums[..., 0], sums[..., 1]              # [N][C]: sum dz, sum dz*xhat
    if film is None:
        return s2.sum(0), s1.sum(0), None
    B = N // T
    sc1 = 1.0 + film[:, :C].repeat_interleave(T, dim=0)   # (1 + scale) per (n, c)
    dscale = (s2 * gamma + s1 * beta).view(B, T, C).sum(1)
    dshift = s1.view(B, T, C).sum(1)
    return (s2 * sc1).sum(0), (s1 * sc1).sum(0), th.cat([dscale, dshift], dim=1)


# ----------------------------------------------------------------------------- ResBlock
dropout_log = None      # tests set this to a list to record the keep masks drawn by ResBlockFn


def _dropout_keep(act, p):
    """Inverted-dropout factors for nn.Dropout(p) in training mode (unet.py:166): 0 or 1/(1-p) per element."""
    keep = (th.rand_like(act) >= p).to(act.dtype).mul_(1.0 / (1.0 - p))
    if dropout_log is not None:
        dropout_log.append(keep)
    return keep


class ResBlockFn(th.autograd.Function):
    """reference unet.py:194-207 with use_scale_shift_norm=True, on a virtual concat input (a | b)."""

    @staticmethod
    def forward(ctx, a, b, film, g1, be1, w1, b1, g2, be2, w2, b2, ws, bs, N, H, W, T, dfilm_slot=None, drop_p=0.0,
                a_slot=None, b_slot=None):
        """a_slot / b_slot (``_SkipSlot``, in-place mode): a_slot holds a further gradient of ``a`` to be added to this
        node's; b_slot receives this node's gradient of ``b`` instead of autograd."""
        C0 = a.shape[1]
        C1 = b.shape[1] if b is not None else 0
        Cin, Cout, P = C0 + C1, w1.shape[0], H * W
        act1, cA1, cB1, st1 = _gn_apply(a, b, C0, C1, N, P, g1, be1, None, T, nat.ACT_SILU)
        h1 = _new(N * P, Cout, like=a)
        nat.conv_igemm(src0=act1, C0=Cin, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=_pack(w1), bias=b1, Cout=Cout, out=h1, ldo=Cout)
        act2, cA2, cB2, st2 = _gn_apply(h1, None, Cout, 0, N, P, g2, be2, film, T, nat.ACT_SILU)
        keep = None
        if drop_p > 0.0:
            keep = _dropout_keep(act2, drop_p)
            act2.mul_(keep)
        out = _new(N * P, Cout, like=a)
        kw = dict(src0=act2, C0=Cout, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=_pack(w2), bias=b2, Cout=Cout, out=out, ldo=Cout)
        if ws is None:
            kw.update(res=a, ldr=Cout)
        else:
            kw.update(s2src0=a, s2src1=b, s2C0=C0, s2C1=C1, W2=ws.view(Cout, Cin), bias2=bs)
        nat.conv_igemm(**kw)
        ctx.save_for_backward(a, b, film, g1, be1, w1, g2, be2, w2, ws, h1, cA1, cB1, st1, cA2, cB2, st2, act1, act2, keep)
        ctx.params = (w1, b1, w2, b2, ws, bs)
        ctx.inplace = _leaf(g1, be1, w1, b1, g2, be2, w2, b2, ws, bs)
        assert ctx.inplace or (dfilm_slot is None and a_slot is None and b_slot is None)
        ctx.dfilm_slot = dfilm_slot
        ctx.a_slot, ctx.b_slot = a_slot, b_slot
        ctx.geom = (N, H, W, T, C0, C1, Cout)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, film, g1, be1, w1, g2, be2, w2, ws, h1, cA1, cB1, st1, cA2, cB2, st2, act1, act2, keep = ctx.saved_tensors
        N, H, W, T, C0, C1, Cout = ctx.geom
        Cin, P = C0 + C1, H * W
        dout = dout.contiguous()
        geo = dict(N=N, Hs=H, Ws=W, Ho=H, Wo=W)
        pw1, pb1, pw2, pb2, pws, pbs = ctx.params
        inplace = ctx.inplace

        def wgrad(pw, pb, **kw):        # -> (dW, db) for autograd, or (None, None) after accumulating in place
            if inplace:
                _wgrad_accumulate(pw, pb, **kw)
                return None, None
            return _wgrad_into(tuple(pw.shape), a, **kw)

        # conv2: weight grad on the materialised operand act(GN2(h1)), data grad -> da2
        dw2, db2 = wgrad(pw2, pb2, src0=act2, C0=Cout, res=dout, ldr=Cout, **geo)
        da2 = _new(N * P, Cout, like=a)
        nat.conv_igemm(src0=dout, C0=Cout, W=_pack_t(w2), Cout=Cout, out=da2, ldo=Cout, **geo)
        if keep is not None:
            da2.mul_(keep)
        if ctx.dfilm_slot is not None:
            _embed.ensure_backward_queued()
        dh1, _, dg2, dbe2, dfilm = _gn_backward(da2, h1, None, Cout, 0, N, P, cA2, cB2, st2, nat.ACT_SILU, g2, be2, film, T,
                                                dfilm_out=ctx.dfilm_slot, inplace=inplace)
        # conv1
        dw1, db1 = wgrad(pw1, pb1, src0=act1, C0=Cin, res=dh1, ldr=Cout, **geo)
        da1 = _new(N * P, Cin, like=a)
        nat.conv_igemm(src0=dh1, C0=Cout, W=_pack_t(w1), Cout=Cin, out=da1, ldo=Cin, **geo)
        # skip path first: its gradient w.r.t. the block input is folded into the GroupNorm-1 backward launch
        dws = dbs = None
        extra = ctx.a_slot.take() if ctx.a_slot is not None else None      # the decoder's gradient of `a` (C1 == 0 then)
        if ws is None:
            skip_grad = dout                       # identity skip (C1 == 0, Cin == Cout)
        else:
            dws, dbs = wgrad(pws, pbs, src0=a, src1=b, C0=C0, C1=C1, ksize=1, res=dout, ldr=Cout, **geo)
            skip_grad = _new(N * P, Cin, like=a)
            rkw = {}
            if extra is not None:                  # rides in the epilogue of the skip path's data-gradient GEMM
                rkw, extra = dict(res=extra, ldr=Cin), None
            nat.conv_igemm(src0=dout, C0=Cout, ksize=1, W=_pack_t(ws), Cout=Cin, out=skip_grad, ldo=Cin, **geo, **rkw)
        dxa, dxb, dg1, dbe1, _ = _gn_backward(da1, a, b, C0, C1, N, P, cA1, cB1, st1, nat.ACT_SILU, g1, be1, None, T,
                                              inplace=inplace, add=skip_grad, add2=extra)
        if ctx.b_slot is not None:
            ctx.b_slot.put(dxb)
            dxb = None
        return (dxa, dxb, dfilm, dg1, dbe1, dw1, db1, dg2, dbe2, dw2, db2, dws, dbs, None, None, None, None, None, None,
                None, None)


# ----------------------------------------------------------------------------- output head
_head_rows = {}      # (rows, channels, device) -> zero-padded gradient rows of the output head (in-place mode)


class HeadFn(th.autograd.Function):
    """out = conv3x3(SiLU(GN(h))) written in the (B,T,C,H,W) frame layout (reference unet.py:399-403,462-464)."""

    @staticmethod
    def forward(ctx, h, g, be, w, b, N, H, W):
        C, Cout, P = h.shape[1], w.shape[0], H * W
        act, cA, cB, st = _gn_apply(h, None, C, 0, N, P, g, be, None, 1, nat.ACT_SILU)
        out = _new(N, Cout, H, W, like=h)
        nat.conv_igemm(src0=act, C0=C, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=_pack(w), bias=b, Cout=Cout, out=out, ldo=Cout,
                       out_mode=nat.OUT_NCHW)
        ctx.save_for_backward(h, g, be, w, cA, cB, st, act)
        ctx.inplace = _leaf(g, be)
        ctx.params = (w, b) if _leaf(w, b) else None
        ctx.geom = (N, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, g, be, w, cA, cB, st, act = ctx.saved_tensors
        N, H, W = ctx.geom
        C, Cout, P = h.shape[1], w.shape[0], H * W
        # rows [M][32]: the data-gradient GEMM reduces over Cout, padded to one 32-channel chunk
        CP = (Cout + 31) // 32 * 32
        geo = dict(N=N, Hs=H, Ws=W, Ho=H, Wo=W)
        if ctx.params is not None:
            # in-place mode: persistent zero-padded rows (only the real channels are rewritten), the weight gradient
            # accumulated like any other 3x3 layer's, the data-gradient operand from the padded pack of the parameter
            key = (N * P, CP, h.device)
            drows = _head_rows.get(key)
            if drows is None:
                drows = _head_rows[key] = th.zeros(N * P, CP, device=h.device, dtype=th.float32)
            drows.view(N, P, CP)[:, :, :Cout].copy_(dout.reshape(N, Cout, P).permute(0, 2, 1))
            _wgrad_accumulate(ctx.params[0], ctx.params[1], src0=act, C0=C, res=drows, ldr=CP, **geo)
            da = _new(N * P, C, like=h)
            nat.conv_igemm(src0=drows, C0=CP, W=_pack_t(w, CP), Cout=C, out=da, ldo=C, **geo)
            dh, _, dg, dbe, _ = _gn_backward(da, h, None, C, 0, N, P, cA, cB, st, nat.ACT_SILU, g, be, None, 1, inplace=ctx.inplace)
            return dh, dg, dbe, None, None, None, None, None
        drows = th.zeros(N * P, CP, device=h.device, dtype=th.float32)
        drows[:, :Cout] = dout.permute(0, 2, 3, 1).reshape(N * P, Cout)
        gp = th.zeros(CP, 9, C, device=h.device, dtype=th.float32)
        dbp = th.zeros(CP, device=h.device, dtype=th.float32)
        nat.conv_wgrad(src0=act, C0=C, res=drows, ldr=CP, out=gp, bias=dbp, Cout=CP, **geo)
        dw = gp[:Cout].view(Cout, 3, 3, C).permute(0, 3, 1, 2).contiguous()
        wt = th.zeros(C, 9, CP, device=h.device, dtype=th.float32)      # Wt[ci][t][co] = W[co][ci][8-t]
        wt[:, :, :Cout] = w.flip(2, 3).reshape(Cout, C, 9).permute(1, 2, 0)
        da = _new(N * P, C, like=h)
        nat.conv_igemm(src0=drows, C0=CP, W=wt, Cout=C, out=da, ldo=C, **geo)
        dh, _, dg, dbe, _ = _gn_backward(da, h, None, C, 0, N, P, cA, cB, st, nat.ACT_SILU, g, be, None, 1, inplace=ctx.inplace)
        return dh, dg, dbe, dw, dbp[:Cout].clone(), None, None, None


# ----------------------------------------------------------------------------- attention
class TemporalAttnFn(th.autograd.Function):
    """x -> GN_t(x) + proj(attn_rpe(qkv(GN_t(x))))   (reference rpe.py:133-174, temporal instance)."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, wqkv, bqkv, wproj, bproj, Rq, Rk, Rv, mask, B, T, P, heads, dR_slots=None, attn_sink=None):
        C = x.shape[1]
        M = B * T * P
        xn = th.empty_like(x)
        nat.gn_temporal(x, gn_w, gn_b, _EPS, xn, B, T, P, C)
        qkv = _new(M, 3 * C, like=x)
        nat.conv_igemm(src0=xn, C0=C, N=M, Hs=1, Ws=1, Ho=1, Wo=1, ksize=1, W=wqkv, bias=bqkv, Cout=3 * C, out=qkv, ldo=3 * C)
        o = _new(M, C, like=x)
        probs = _new(B * P, heads, T, T, like=x) if attn_sink is not None else None
        nat.attn_temporal(qkv, Rq, Rk, Rv, mask, o, probs, B, T, P, C, heads)
        if probs is not None:       # logged summary of the reference: |mean over heads|, detached (rpe.py:128-131)
            attn_sink.append(probs.mean(dim=1).abs())
        y = _new(M, C, like=x)
        nat.conv_igemm(src0=o, C0=C, N=M, Hs=1, Ws=1, Ho=1, Wo=1, ksize=1, W=wproj, bias=bproj, Cout=C, res=xn, ldr=C, out=y,
                       ldo=C)
        ctx.save_for_backward(x, gn_w, wqkv, wproj, Rq, Rk, Rv, mask, xn, qkv, o)
        ctx.params = (wqkv, bqkv, wproj, bproj)
        ctx.gn_b = gn_b
        ctx.inplace = _leaf(gn_w, gn_b, wqkv, bqkv, wproj, bproj)
        ctx.dR_slots = dR_slots        # grouped RPE path: dR_q/k/v go to the group's buffers, not to autograd
        ctx.geom = (B, T, P, heads)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gn_w, wqkv, wproj, Rq, Rk, Rv, mask, xn, qkv, o = ctx.saved_tensors
        B, T, P, heads = ctx.geom
        C = x.shape[1]
        M = B * T * P
        dy = dy.contiguous()
        one = dict(N=M, Hs=1, Ws=1, Ho=1, Wo=1)
        pwq, pbq, pwp, pbp = ctx.params
        inplace = ctx.inplace

        def wgrad(pw, pb, **kw):
            if inplace:
                _wgrad_accumulate(pw, pb, **kw)
                return None, None
            return _wgrad_into(tuple(pw.shape), x, **kw)

        dwp, dbp = wgrad(pwp, pbp, src0=o, C0=C, res=dy, ldr=C, **one)
        do = _new(M, C, like=x)
        nat.conv_igemm(src0=dy, C0=C, ksize=1, W=_pack_t(wproj.view(C, C, 1, 1)), Cout=C, out=do, ldo=C, **one)
        # attention core backward: HIP kernels (rows -> dq, P, dS; cols -> dk, dv; rpe -> dR_q/k/v over pixels)
        dqkv = _new(M, 3 * C, like=x)
        if ctx.dR_slots is not None:
            dRq, dRk, dRv = ctx.dR_slots
            _rpe_group.pending = True
            _embed.ensure_backward_queued()          # its callback runs the grouped RPE backward first
        else:
            dRq, dRk, dRv = (_new(B * T * T, C, like=x).view(B, T, T, C) for _ in range(3))
        ws_p, ws_ds = _new(B * P * heads * T, T, like=x), _new(B * P * heads * T, T, like=x)
        nat.attn_temporal_bwd(qkv, do, Rq, Rk, Rv, mask, ws_p, ws_ds, dqkv, dRq, dRk, dRv, B, T, P, C, heads)
        dwq, dbq = wgrad(pwq, pbq, src0=xn, C0=C, res=dqkv, ldr=3 * C, **one)
        dxn = _new(M, C, like=x)
        nat.conv_igemm(src0=dqkv, C0=3 * C, ksize=1, W=_pack_t(wqkv.view(3 * C, C, 1, 1)), Cout=C, res=dy, ldr=C, out=dxn,
                       ldo=C, **one)
        dx = th.empty_like(x)
        gn_b = ctx.gn_b
        if inplace:      # accumulate straight into the parameter gradients
            dg, db, tg, tb = None, None, _grad_of(gn_w), _grad_of(gn_b)
        else:
            dg, db = th.zeros(C, device=x.device), th.zeros(C, device=x.device)
            tg, tb = dg, db
        if nat.deterministic():
            ws = nat.det_workspace(x.device)
            nat.check(nat.lib().lfvdm_gn_temporal_bwd_det(nat.ptr(x), nat.ptr(dxn), nat.ptr(gn_w), _EPS, nat.ptr(dx), nat.ptr(tg),
                                                          nat.ptr(tb), B, T, P, C, 0, ws.data_ptr(), ws.numel(), nat.stream()),
                      "lfvdm_gn_temporal_bwd_det")
        else:
            nat.check(nat.lib().lfvdm_gn_temporal_bwd(nat.ptr(x), nat.ptr(dxn), nat.ptr(gn_w), _EPS, nat.ptr(dx), nat.ptr(tg),
                                                      nat.ptr(tb), B, T, P, C, 0, nat.stream()), "lfvdm_gn_temporal_bwd")
        if ctx.dR_slots is not None:
            dRq = dRk = dRv = None
        return dx, dg, db, dwq, dbq, dwp, dbp, dRq, dRk, dRv, None, None, None, None, None, None, None


class SpatialAttnFn(th.autograd.Function):
    """x -> GN_s(x) + proj(attn(qkv(GN_s(x)))) over the H*W tokens of each frame (rpe.py:133-174, spatial)."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, wqkv, bqkv, wproj, bproj, N, P, heads, attn_sink=None):
        C = x.shape[1]
        M = N * P
        xn, cA, cB, st = _gn_apply(x, None, C, 0, N, P, gn_w, gn_b, None, 1, nat.ACT_NONE)   # also the residual
        qkv = _new(M, 3 * C, like=x)
        nat.conv_igemm(src0=xn, C0=C, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=wqkv, bias=bqkv, Cout=3 * C, out=qkv, ldo=3 * C)
        o = _new(M, C, like=x)
        lse = _new(N * heads, P, like=x)
        probs = _new(N, heads, P, P, like=x) if attn_sink is not None else None
        nat.attn_spatial(qkv, o, probs, N, P, C, heads, lse=lse)
        if probs is not None:
            attn_sink.append(probs.mean(dim=1).abs())
        y = _new(M, C, like=x)
        nat.conv_igemm(src0=o, C0=C, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=wproj, bias=bproj, Cout=C, res=xn, ldr=C, out=y,
                       ldo=C)
        ctx.save_for_backward(x, gn_w, gn_b, wqkv, wproj, cA, cB, st, qkv, o, lse, xn)
        ctx.params = (wqkv, bqkv, wproj, bproj)
        ctx.inplace = _leaf(gn_w, gn_b, wqkv, bqkv, wproj, bproj)
        ctx.geom = (N, P, heads)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gn_w, gn_b, wqkv, wproj, cA, cB, st, qkv, o, lse, xn = ctx.saved_tensors
        N, P, heads = ctx.geom
        C = x.shape[1]
        M, Fh = N * P, C // heads
        dy = dy.contiguous()
        geo = dict(N=N, Hs=P, Ws=1, Ho=P, Wo=1)
        pwq, pbq, pwp, pbp = ctx.params
        inplace = ctx.inplace

        def wgrad(pw, pb, **kw):
            if inplace:
                _wgrad_accumulate(pw, pb, **kw)
                return None, None
            return _wgrad_into(tuple(pw.shape), x, **kw)

        dwp, dbp = wgrad(pwp, pbp, src0=o, C0=C, ksize=1, res=dy, ldr=C, **geo)
        do = _new(M, C, like=x)
        nat.conv_igemm(src0=dy, C0=C, ksize=1, W=_pack_t(wproj.view(C, C, 1, 1)), Cout=C, out=do, ldo=C, **geo)
        # core backward: flash-style HIP kernels (S, P recomputed from q, k and the saved log-sum-exp)
        dqkv = _new(M, 3 * C, like=x)
        nat.attn_spatial_bwd(qkv, o, do, lse, _new(N * heads, P, like=x), dqkv, N, P, C, heads)
        dwq, dbq = wgrad(pwq, pbq, src0=xn, C0=C, ksize=1, res=dqkv, ldr=3 * C, **geo)
        dxn = _new(M, C, like=x)   # gradient w.r.t. the normalised tensor: qkv path + residual
        nat.conv_igemm(src0=dqkv, C0=3 * C, ksize=1, W=_pack_t(wqkv.view(3 * C, C, 1, 1)), Cout=C, res=dy, ldr=C, out=dxn,
                       ldo=C, **geo)
        dx, _, dg, db, _ = _gn_backward(dxn, x, None, C, 0, N, P, cA, cB, st, nat.ACT_NONE, gn_w, gn_b, None, 1, inplace=inplace)
        return dx, dg, db, dwq, dbq, dwp, dbp, None, None, None, None


# ----------------------------------------------------------------------------- embedding network
class _EmbedNet:
    """Everything that depends on the diffusion timestep only - sinusoid, time_embed MLP (unet.py:303-308), the
    FiLM projection of every ResBlock (unet.py:157-163) and embed_diffusion_time of every RPE network
    (rpe.py:29) - as THREE grouped launches forward and three backward, differentiated by hand outside the
    autograd tape.  Consumers (ResBlockFn / RpeFrontFn) read strided views of one flat output buffer and write the
    gradients of those views into the matching slots of one flat gradient buffer; the backward runs once, as an
    end-of-backward callback, and accumulates straight into the parameters' ``.grad``."""

    def __init__(self):
        self.state = None
        self.queued = False

    def _build(self, m, B, dev):
        from .unet import ResBlock, FactorizedAttentionBlock
        ch = m.model_channels
        ted = 4 * ch
        half = ch // 2
        Bp = (B + 3) // 4 * 4
        lin0, lin1 = m.time_embed[0], m.time_embed[2]
        heads = []         # (module key, weight, bias, in_mode)
        for blk in list(m.input_blocks) + [m.middle_block] + list(m.output_blocks):
            for layer in blk:
                if isinstance(layer, ResBlock):
                    heads.append((layer, layer.emb_layers[1].weight, layer.emb_layers[1].bias, 1))
                elif isinstance(layer, FactorizedAttentionBlock):
                    ta = layer.temporal_attention
                    for r in (ta.rpe_q, ta.rpe_k, ta.rpe_v):
                        lin = r.rpe_net.embed_diffusion_time
                        heads.append((r.rpe_net, lin.weight, lin.bias, 0))
        params = [lin0.weight, lin0.bias, lin1.weight, lin1.bias] + [t for _, w, b, _ in heads for t in (w, b)]
        if not all(p.is_leaf for p in params):
            return None
        total = sum(w.shape[0] for _, w, _, _ in heads)
        z = lambda *shape: th.zeros(*shape, device=dev, dtype=th.float32)
        st = dict(B=B, dev=dev, params=params, tin=z(Bp + half), h0=z(B, ted), emb=z(B, ted), flat=z(B, total),
                  grads=z(B, total + 3 * ted), total=total, ted=ted)
        st["tin"][Bp:] = timestep_freqs(ch).to(dev)
        g = st["grads"]
        st["dflat"], st["demb_act"], st["demb_raw"], st["dh0_act"] = (
            g[:, :total], g[:, total:total + ted], g[:, total + ted:total + 2 * ted], g[:, total + 2 * ted:])
        ld = g.stride(0)
        P = lambda t: t.data_ptr()
        G = lambda p: _grad_of(p).data_ptr()
        fwd0 = [nat.RowdotJob(P(lin0.weight), P(lin0.bias), P(st["tin"]), P(st["h0"]), ch, ted, B, Bp, ted, 2, 0, 0)]
        fwd1 = [nat.RowdotJob(P(lin1.weight), P(lin1.bias), P(st["h0"]), P(st["emb"]), ted, ted, B, ted, ted, 1, 0, 0)]
        fwdg, bwdg, views, off, task0 = [], [], {}, 0, 0
        for key, w, b, mode in heads:
            O = w.shape[0]
            fwdg.append(nat.RowdotJob(P(w), P(b), P(st["emb"]), P(st["flat"]) + 4 * off, ted, O, B, ted, total, mode, off, 0))
            din = st["demb_act"] if mode == 1 else st["demb_raw"]
            bwdg.append(nat.RowdotBwdJob(P(w), P(st["emb"]), P(st["dflat"]) + 4 * off, G(w), G(b), P(din), ted, O, B, ted, ld, ld,
                                         mode, task0))
            views[key] = (st["flat"][:, off:off + O], st["dflat"][:, off:off + O])
            off += O
            task0 += (O + 7) // 8
        st["demb"] = z(B, ted)
        st["dh0"] = z(B, ted)
        bwd1 = [nat.RowdotBwdJob(P(lin1.weight), P(st["h0"]), P(st["demb"]), G(lin1.weight), G(lin1.bias), P(st["dh0_act"]), ted,
                                 ted, B, ted, ted, ld, 1, 0)]
        bwd0 = [nat.RowdotBwdJob(P(lin0.weight), P(st["tin"]), P(st["dh0"]), G(lin0.weight), G(lin0.bias), None, ch, ted, B, Bp,
                                 ted, 0, 2, 0)]
        J = lambda jobs: nat.jobs_to_device(jobs, dev)
        st.update(views=views, j_f0=J(fwd0), j_f1=J(fwd1), j_fg=J(fwdg), n_g=len(fwdg), j_bg=J(bwdg), tasks_g=task0, j_b1=J(bwd1),
                  j_b0=J(bwd0), tasks_t=(ted + 7) // 8,
                  key=(id(m), B, str(dev), tuple(p.data_ptr() for p in params), tuple(_grad_of(p).data_ptr() for p in params)))
        return st

    def forward(self, m, timesteps, B, dev):
        """-> {module: (output view, gradient slot)} or None if the grouped path does not apply."""
        if not _mode.inplace:
            return None                 # gradients must flow through autograd: per-layer path
        st = self.state
        if st is not None:
            ps = st["params"]
            if st["key"] != (id(m), B, str(dev), tuple(p.data_ptr() for p in ps),
                             tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in ps)):
                st = None
        if st is None:
            if th.cuda.is_current_stream_capturing():
                return None             # job tables cannot be uploaded now: per-layer path
            st = self.state = self._build(m, B, dev)
            if st is None:
                return None
        L, s = nat.lib(), nat.stream()
        st["tin"][:B].copy_(timesteps)
        st["grads"].zero_()
        nat.check(L.lfvdm_rowdot(st["j_f0"].data_ptr(), 1, st["ted"], s), "lfvdm_rowdot")
        nat.check(L.lfvdm_rowdot(st["j_f1"].data_ptr(), 1, st["ted"], s), "lfvdm_rowdot")
        nat.check(L.lfvdm_rowdot(st["j_fg"].data_ptr(), st["n_g"], st["total"], s), "lfvdm_rowdot")
        return st["views"]

    def ensure_backward_queued(self):
        if not self.queued:
            self.queued = True
            th.autograd.Variable._execution_engine.queue_callback(self.backward)

    def backward(self):
        self.queued = False
        st = self.state
        if _rpe_group.pending:          # fills the RPE projections' gradient slots of this buffer
            _rpe_group.backward()
        L, s = nat.lib(), nat.stream()
        if nat.deterministic():
            # every `din` of these launches lives in the flat gradient buffer st["grads"]: ordered slabs over it
            ws, g = nat.det_workspace(st["dev"]), st["grads"]

            def rowdot_bwd(jobs, n, tasks, has_din=True):
                nat.check(L.lfvdm_rowdot_bwd_det(jobs.data_ptr(), n, tasks, g.data_ptr() if has_din else None, g.numel(), ws.data_ptr(),
                                                 ws.numel(), s), "lfvdm_rowdot_bwd_det")
        else:
            def rowdot_bwd(jobs, n, tasks, has_din=True):
                nat.check(L.lfvdm_rowdot_bwd(jobs.data_ptr(), n, tasks, s), "lfvdm_rowdot_bwd")
        rowdot_bwd(st["j_bg"], st["n_g"], st["tasks_g"])
        # d emb = (through the RPE projections) + (through silu in front of the FiLM projections)
        th.add(st["demb_raw"], th.ops.aten.silu_backward(st["demb_act"], st["emb"]), out=st["demb"])
        rowdot_bwd(st["j_b1"], 1, st["tasks_t"])
        st["dh0"].copy_(th.ops.aten.silu_backward(st["dh0_act"], st["h0"]))
        rowdot_bwd(st["j_b0"], 1, st["tasks_t"], has_din=False)


_embed = _EmbedNet()


class _RpeGroup:
    """All RPE networks of a training step (rpe.py:20-31; 3 per temporal attention, 21 at the reference depth) as
    grouped launches, in-place mode only: ONE lfvdm_rpe_nets launch forward (hidden layer + output layer, storing the
    hidden activations), and two at the end of the backward pass - lfvdm_rpe_nets_bwd (data gradient of the output
    layer + backward of the hidden layer, into the embedding network's gradient slots) and lfvdm_conv_wgrad_grouped
    (the output layers' weight / bias gradients).  Per network that replaces ~6 small launches (~35 us).  The R
    tensors, their gradients and the activations live in static buffers owned by the group, so the device job tables
    are built once; TemporalAttnFn writes dR_q/k/v straight into the group's buffers."""

    def __init__(self):
        self.state = None
        self.pending = False

    @staticmethod
    def _nets(m):
        from .unet import FactorizedAttentionBlock
        out = []
        for blk in list(m.input_blocks) + [m.middle_block] + list(m.output_blocks):
            for layer in blk:
                if isinstance(layer, FactorizedAttentionBlock):
                    ta = layer.temporal_attention
                    out.append((layer, [r.rpe_net for r in (ta.rpe_q, ta.rpe_k, ta.rpe_v)]))
        return out

    def _build(self, m, views, B, T, dev):
        nets = self._nets(m)
        flat = [n for _, trio in nets for n in trio]
        if not flat or T * T < 32:
            return None
        params = [t for n in flat for t in (n.embed_distances.weight, n.embed_distances.bias, n.out.weight, n.out.bias)]
        if not all(p.is_leaf for p in params) or any(n.out.weight.shape[0] % 32 or n.out.weight.shape[0] > 512 for n in flat):
            return None
        M = B * T * T
        tiles = (M + 31) // 32
        msplit = max(1, min(4, tiles // 2))
        z = lambda *shape: th.zeros(*shape, device=dev, dtype=th.float32)
        st = dict(bufs={}, keep=[])
        fj, bj, wj, wargs, tile0, task0 = [], [], [], [], 0, 0
        for net in flat:
            C = net.out.weight.shape[0]
            tproj, dtproj = views[net]
            R, dR, act = z(B, T, T, C), z(B, T, T, C), z(M, C)
            wt = _pack_t(net.out.weight.view(C, C, 1, 1))
            st["bufs"][net] = (R, dR)
            st["keep"] += [act, wt]
            fj.append(nat.RpeJob(tproj.data_ptr(), nat.ptr(net.embed_distances.weight), nat.ptr(net.embed_distances.bias),
                                 nat.ptr(net.out.weight), nat.ptr(net.out.bias), nat.ptr(R), C, tile0, tproj.stride(0), 0,
                                 nat.ptr(act)))
            bj.append(nat.RpeBwdJob(tproj.data_ptr(), nat.ptr(net.embed_distances.weight), nat.ptr(net.embed_distances.bias),
                                    nat.ptr(wt), nat.ptr(dR), dtproj.data_ptr(), nat.ptr(_grad_of(net.embed_distances.weight)),
                                    nat.ptr(_grad_of(net.embed_distances.bias)), C, tile0, tproj.stride(0), dtproj.stride(0)))
            a = nat.fill_conv_args(src0=act, C0=C, N=M, Hs=1, Ws=1, Ho=1, Wo=1, ksize=1, res=dR.view(M, C), ldr=C,
                                   out=_grad_of(net.out.weight), bias=_grad_of(net.out.bias), Cout=C)
            wj.append(nat.WgradJob(a, msplit, task0))
            wargs.append(a)
            tile0 += tiles
            task0 += (C // 32) * (C // 32) * msplit
        st.update(nets=nets, flat=flat, params=params, tiles=tile0, tasks=task0, n=len(flat), wargs=wargs,
                  maxC=max(n.out.weight.shape[0] for n in flat),
                  j_f=nat.jobs_to_device(fj, dev), j_b=nat.jobs_to_device(bj, dev), j_w=nat.jobs_to_device(wj, dev),
                  key=self._key(m, views, flat, params, B, T, dev))
        return st

    @staticmethod
    def _key(m, views, flat, params, B, T, dev):
        if any(n not in views for n in flat):      # tables of another model
            return None
        return (id(m), B, T, str(dev), tuple(p.data_ptr() for p in params),
                tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in params),
                tuple(views[n][0].data_ptr() for n in flat))

    def forward(self, m, views, frame_indices, B, T, dev):
        """-> {attention block: ([R_q, R_k, R_v], [dR_q, dR_k, dR_v])} or None if the grouped path does not apply."""
        if not _mode.inplace or os.environ.get("LFVDM_RPE_GROUPED", "1") == "0":
            return None
        st = self.state
        if st is not None and st["key"] != self._key(m, views, st["flat"], st["params"], B, T, dev):
            st = None
        if st is None:
            if th.cuda.is_current_stream_capturing():
                return None             # job tables cannot be uploaded now: per-network path
            st = self.state = self._build(m, views, B, T, dev)
            if st is None:
                return None
        fi = frame_indices.to(th.int64).contiguous()
        st["fi"] = fi
        nat.check(nat.lib().lfvdm_rpe_nets_maxc(st["j_f"].data_ptr(), st["n"], st["tiles"], nat.ptr(fi, th.int64), B, T,
                                                st["maxC"], nat.stream()), "lfvdm_rpe_nets")
        st["BT"] = (B, T)
        return {layer: ([st["bufs"][n][0] for n in trio], [st["bufs"][n][1] for n in trio]) for layer, trio in st["nets"]}

    def backward(self):
        self.pending = False
        st = self.state
        B, T = st["BT"]
        L, s = nat.lib(), nat.stream()
        if nat.deterministic():
            ws = nat.det_workspace(st["fi"].device)
            nat.check(L.lfvdm_rpe_nets_bwd_det(st["j_b"].data_ptr(), st["n"], st["tiles"], nat.ptr(st["fi"], th.int64), B, T,
                                               ws.data_ptr(), ws.numel(), s), "lfvdm_rpe_nets_bwd_det")
            for a in st["wargs"]:        # the output layers' weight gradients one by one (each through its ordered slab)
                a.splitk_ws, a.splitk_ws_floats = ws.data_ptr(), ws.numel()
                nat.check(L.lfvdm_conv_wgrad(ctypes.byref(a), s), "lfvdm_conv_wgrad")
            return
        nat.check(L.lfvdm_rpe_nets_bwd(st["j_b"].data_ptr(), st["n"], st["tiles"], nat.ptr(st["fi"], th.int64), B, T, s),
                  "lfvdm_rpe_nets_bwd")
        nat.check(L.lfvdm_conv_wgrad_grouped(st["j_w"].data_ptr(), st["n"], st["tasks"], s), "lfvdm_conv_wgrad_grouped")


_rpe_group = _RpeGroup()


# ----------------------------------------------------------------------------- whole network
def _rpe_feats(rel):
    """Distance features of rpe.py:22-27, shared by all RPE networks of a forward pass."""
    relf = rel.to(th.float32)
    return th.stack([th.log1p(relf.clamp(min=0)), th.log1p((-relf).clamp(min=0)), (rel == 0).to(th.float32)], dim=-1).contiguous()


class RpeFrontFn(th.autograd.Function):
    """act = silu(tproj[b] + embed_distances(feats)) on rows (b, t, s): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, tproj, feats, wd, bd, B, TT, dtproj_slot=None):
        C = tproj.shape[1]
        act = _new(B * TT, C, like=tproj)
        nat.check(nat.lib().lfvdm_rpe_front(tproj.data_ptr(), tproj.stride(0), nat.ptr(feats), nat.ptr(wd), nat.ptr(bd),
                                            nat.ptr(act), B, TT, C, nat.stream()), "lfvdm_rpe_front")
        ctx.save_for_backward(tproj, feats, wd, bd)
        ctx.slot = dtproj_slot
        ctx.inplace = _leaf(wd, bd)
        assert ctx.inplace or dtproj_slot is None
        ctx.geom = (B, TT, C)
        return act

    @staticmethod
    def backward(ctx, d_act):
        tproj, feats, wd, bd = ctx.saved_tensors
        B, TT, C = ctx.geom
        leaf = ctx.inplace
        if ctx.slot is not None:      # accumulate into the embedding network's (zeroed) gradient buffer
            _embed.ensure_backward_queued()
            dtproj = ctx.slot
        else:
            dtproj = th.zeros(B, C, device=tproj.device, dtype=th.float32)
        dwd = _grad_of(wd) if leaf else th.zeros_like(wd)
        dbd = _grad_of(bd) if leaf else th.zeros_like(bd)
        nat.check(nat.lib().lfvdm_rpe_front_bwd(tproj.data_ptr(), tproj.stride(0), nat.ptr(feats), nat.ptr(wd), nat.ptr(bd),
                                                nat.ptr(d_act.contiguous()), dtproj.data_ptr(), dtproj.stride(0), nat.ptr(dwd),
                                                nat.ptr(dbd), B, TT, C, nat.stream()), "lfvdm_rpe_front_bwd")
        return (None if ctx.slot is not None else dtproj), None, (None if leaf else dwd), (None if leaf else dbd), None, None, None


def _rpe_R(net, temb_b, feats, B, T, tproj=None, slot=None):
    """RPENet (rpe.py:20-31) on device: the 3-feature / time-embedding projections are tiny library ops, the
    C x C output layer runs on the HIP GEMM kernels."""
    C = net.out.weight.shape[0]
    if C % 32 == 0:   # hidden layer fused in one launch; C x C output layer on the GEMM / wgrad kernels (rows = B*T*T)
        if tproj is None:
            tproj = net.embed_diffusion_time(temb_b)
        act = RpeFrontFn.apply(tproj, feats.view(B * T * T, 3), net.embed_distances.weight, net.embed_distances.bias, B, T * T,
                               slot)
        return LinearFn.apply(act, net.out.weight, net.out.bias, None).view(B, T, T, C)
    hid = net.embed_diffusion_time(temb_b).view(B, 1, 1, -1) + net.embed_distances(feats)
    return net.out(F.silu(hid)).contiguous()       # B, T, T, C


class UNetFunction:
    """Differentiable forward of the whole U-Net in training mode (same math as ``_engine.Plan``)."""

    @staticmethod
    def run(engine, x, x0, timesteps, frame_indices, obs_mask, latent_mask, return_attn_weights):
        from .unet import ResBlock, FactorizedAttentionBlock, Downsample, Upsample
        m = engine.model
        attns = {"spatial": [], "temporal": [], "mixed": []} if return_attn_weights else None
        _SkipSlot.reset_stale()
        _mode.inplace = bool(getattr(m, "native_grad_accumulation", False))
        B, T, Cx, H, W = x.shape
        N = B * T
        ch = m.model_channels
        obs = obs_mask.reshape(B, T, 1, 1, 1).to(th.float32)
        mask = (obs_mask.reshape(B, T) + latent_mask.reshape(B, T)).clamp(max=1).to(th.float32).contiguous()
        # --- embeddings (per batch element: rows of the reference's (B*T, 4ch) emb are equal within b)
        views = _embed.forward(m, timesteps.to(th.float32), B, x.device) if x.is_cuda else None
        if views is None:        # per-layer library path (non-leaf parameters, or first call under capture)
            freqs = getattr(engine, "_freqs", None)
            if freqs is None or freqs.device != x.device:
                freqs = engine._freqs = timestep_freqs(ch).to(x.device)     # uploaded once (graph capture safe)
            args = timesteps.to(th.float32)[:, None] * freqs[None]
            temb = th.cat([th.cos(args), th.sin(args)], dim=-1)
            emb = m.time_embed[2](F.silu(m.time_embed[0](temb)))            # (B, 4ch)
            semb = F.silu(emb)
        else:
            emb = semb = None
        # all RPE networks in one grouped launch (reads the frame indices itself); the per-layer path needs the features
        rpe_grp = _rpe_group.forward(m, views, frame_indices, B, T, x.device) if views is not None else None
        feats = _rpe_feats(frame_indices.unsqueeze(-1) - frame_indices.unsqueeze(-2)) if rpe_grp is None else None
        # --- input compositing + first conv (reference unet.py:441-450): 5 -> 32 zero-padded channels, one launch
        if (x.requires_grad or x0.requires_grad) and th.is_grad_enabled():     # gradients w.r.t. the frames: library ops
            comp = th.cat([x * (1 - obs) + x0 * obs, th.ones_like(x[:, :, :1]) * obs], dim=2)
            rows = th.zeros(N * H * W, 32, device=x.device, dtype=th.float32)
            rows[:, :Cx + 1] = comp.reshape(N, Cx + 1, H, W).permute(0, 2, 3, 1).reshape(N * H * W, Cx + 1)
        else:
            rows = th.empty(N * H * W, 32, device=x.device, dtype=th.float32)
            nat.check(nat.lib().lfvdm_compose_rows(nat.ptr(x.contiguous().float()), nat.ptr(x0.contiguous().float()),
                                                   nat.ptr(obs.reshape(N).contiguous()), nat.ptr(rows), N, Cx, H, W, 32,
                                                   nat.stream()), "lfvdm_compose_rows")
        conv0 = m.input_blocks[0][0]
        if _leaf(conv0.weight, conv0.bias) and not rows.requires_grad:
            # the weight's 5 input channels are packed into a 32-channel operand whose padding stays zero, and the weight
            # gradient is accumulated in the same shape and folded with the others (no pad / slice / add launches)
            h = ConvFn.apply(rows, conv0.weight, conv0.bias, N, H, W, 1, False, 32)
        else:
            w0 = F.pad(conv0.weight, (0, 0, 0, 0, 0, 32 - (Cx + 1)))
            h = ConvFn.apply(rows, w0, conv0.bias, N, H, W, 1, False)
        cur = (h, H, W)
        hs = [cur]

        def stage(blk, cur, skip=None, a_slot=None, b_slot=None):
            h, Hc, Wc = cur
            b = skip
            for layer in blk:
                if isinstance(layer, ResBlock):
                    film, slot = views[layer] if views is not None else (layer.emb_layers[1](semb), None)
                    sk, drop = layer.skip_connection, layer.out_layers[2]
                    ws, bs = (None, None) if isinstance(sk, nn.Identity) else (sk.weight, sk.bias)
                    h = ResBlockFn.apply(h, b, film, layer.in_layers[0].weight, layer.in_layers[0].bias,
                                         layer.in_layers[2].weight, layer.in_layers[2].bias, layer.out_layers[0].weight,
                                         layer.out_layers[0].bias, layer.out_layers[3].weight, layer.out_layers[3].bias,
                                         ws, bs, N, Hc, Wc, T, slot, drop.p if drop.training else 0.0, a_slot, b_slot)
                    b = a_slot = b_slot = None
                elif isinstance(layer, FactorizedAttentionBlock):
                    ta, sa = layer.temporal_attention, layer.spatial_attention
                    if rpe_grp is not None:
                        R, dR_slots = rpe_grp[layer]
                    else:
                        R = [_rpe_R(r.rpe_net, emb, feats, B, T, *(views[r.rpe_net] if views is not None else (None, None)))
                             for r in (ta.rpe_q, ta.rpe_k, ta.rpe_v)]
                        dR_slots = None
                    h = TemporalAttnFn.apply(h, ta.norm.weight, ta.norm.bias, ta.qkv.weight, ta.qkv.bias, ta.proj_out.weight,
                                             ta.proj_out.bias, R[0], R[1], R[2], mask, B, T, Hc * Wc, layer.num_heads,
                                             dR_slots, attns["temporal"] if attns is not None else None)
                    h = SpatialAttnFn.apply(h, sa.norm.weight, sa.norm.bias, sa.qkv.weight, sa.qkv.bias, sa.proj_out.weight,
                                            sa.proj_out.bias, N, Hc * Wc, layer.num_heads,
                                            attns["spatial"] if attns is not None else None)
                elif isinstance(layer, Downsample):
                    h = ConvFn.apply(h, layer.op.weight, layer.op.bias, N, Hc, Wc, 2, False, 0, a_slot)
                    a_slot = None
                    Hc, Wc = Hc // 2, Wc // 2
                elif isinstance(layer, Upsample):
                    h = ConvFn.apply(h, layer.conv.weight, layer.conv.bias, N, Hc, Wc, 1, True)
                    Hc, Wc = 2 * Hc, 2 * Wc
                else:
                    raise NotImplementedError(type(layer))
            return (h, Hc, Wc)

        # data-parallel training: marker nodes at the inputs of the stages where a gradient bucket starts; their
        # backward tells the exchange that the bucket is complete (_exchange.GradExchange)
        xch = getattr(m, "_grad_exchange", None) if _mode.inplace else None

        def enter(cur, key):
            if xch is None:
                return cur
            return (xch.mark(cur[0], key), cur[1], cur[2])

        # in-place mode: the decoder's gradient of every skip tensor goes through a _SkipSlot to the kernel that produces
        # the encoder-side gradient of the same tensor (no autograd sum per skip connection)
        def slot_for(nxt):
            first = nxt[0] if len(nxt) else None
            ok = _mode.inplace and th.is_grad_enabled() and isinstance(first, (ResBlock, Downsample)) and _SKIP_SLOTS
            return _SkipSlot() if ok and all(p.is_leaf for p in first.parameters()) else None

        enc = list(m.input_blocks)[1:]
        slots = [slot_for(enc[0] if enc else m.middle_block)]
        for i, blk in enumerate(enc, start=1):
            cur = stage(blk, enter(cur, (0, i)), a_slot=slots[-1])
            hs.append(cur)
            slots.append(slot_for(enc[i] if i < len(enc) else m.middle_block))
        cur = stage(m.middle_block, enter(cur, (1, 0)), a_slot=slots[-1])
        for i, blk in enumerate(m.output_blocks):
            cur = stage(blk, enter(cur, (2, i)), skip=hs.pop()[0], b_slot=slots.pop())
        cur = enter(cur, (3, 0))
        h, Hc, Wc = cur
        out = HeadFn.apply(h, m.out[0].weight, m.out[0].bias, m.out[2].weight, m.out[2].bias, N, Hc, Wc)
        return out.view(B, T, m.out_channels, H, W), attns
