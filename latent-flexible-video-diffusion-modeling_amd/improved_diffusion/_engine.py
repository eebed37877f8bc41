"""Native execution engine of ``UNetVideoModel``: turns the module tree into a flat sequence of
gfx950 kernel launches (C ABI of include/lfvdm_hip.h) for one input shape.

Design (MI355X-first, not a translation of the reference's ATen op stream, unet.py:428-464):
  * activations are channels-last [B*T][H][W][C]; no layout permutes exist anywhere — temporal
    attention strides over frames, spatial attention over pixels of the same buffer;
  * GroupNorm(+FiLM)+SiLU is evaluated ONCE per use - in the epilogue of the GEMM that produces its input where a
    tile holds whole samples, else by lfvdm_gn_apply - and the implicit GEMMs stage raw operands by LDS-DMA;
  * skip concat, nearest-2x upsample, the 1x1 skip conv and every residual add are operands /
    epilogues of the implicit-GEMM kernel;
  * everything that depends only on (t, frame_indices) — time-embedding MLP, all ResBlock FiLM
    projections, all 3*L RPE networks — is evaluated up front in 4 grouped launches;
  * a forward is ~130 launches with no host synchronisation and no allocation, so the whole
    denoising step is captured into one hipGraph by the sampler (gaussian_diffusion.py).
"""
import ctypes as C
import json
import os
import sys
import weakref

import torch as th
import torch.nn as nn

from . import _native as nat
from .nn import timestep_freqs


def _p(t):
    return t.data_ptr()


# LFVDM_SPATIAL_FUSED=0: keep the qkv projection of the spatial attention as its own GEMM launch (A/B aid)
SPATIAL_FUSED = os.environ.get("LFVDM_SPATIAL_FUSED", "1") != "0"
# LFVDM_PROJ_GN=0: keep the temporal output projection and the spatial GroupNorm of the 16x16 level as two launches (A/B aid)
PROJ_GN = os.environ.get("LFVDM_PROJ_GN", "1") != "0"
# LFVDM_TEMPORAL_QKV=0: keep the temporal GroupNorm and the temporal qkv projection as two launches (A/B aid)
TEMPORAL_QKV = os.environ.get("LFVDM_TEMPORAL_QKV", "1") != "0"
# LFVDM_GN_EPILOGUE=0: never evaluate a GroupNorm in the epilogue of the GEMM that produces its input (A/B aid)
GN_EPILOGUE = os.environ.get("LFVDM_GN_EPILOGUE", "1") != "0"
# LFVDM_NEXT_GN_EPILOGUE=0: never evaluate the NEXT ResBlock's first GroupNorm in a producer's epilogue (A/B aid)
NEXT_GN_EPILOGUE = os.environ.get("LFVDM_NEXT_GN_EPILOGUE", "1") != "0"
# LFVDM_LEVEL_CHAIN=0: every launch of the low-resolution levels stays its own launch (A/B aid; the persistent level chain
# is bitwise the per-launch plan with the same tile codes).  LFVDM_CHAIN_MAX_M: launches with at most this many output rows
# (N * Ho * Wo) are candidates (640 = the 4x4 and 2x2 levels of the 20-frame, batch-2 headline configuration)
def level_chain_enabled():
    """Read when a plan is built / tuned (not at import): a process that learns late that it SHARES its GPU with another
    process - bench.py's N-ranks-on-fewer-cards rehearsal - switches the chains off for the plans it builds afterwards.  A
    chain needs all its workgroups co-resident; two processes' chains on one card can each get part of the CUs and wait for
    workgroups that cannot start (every wait is bounded: the timeout raises the abort word and the sampler falls back, but
    that costs LFVDM_CHAIN_TIMEOUT_S seconds)."""
    return os.environ.get("LFVDM_LEVEL_CHAIN", "1") != "0"


def chain_local_enabled():
    """LFVDM_CHAIN_LOCAL=0: the chains keep the split-K tile body of round 5 for every stage (A/B aid).  Default: stages on
    maps of <= 16 pixels run SAMPLE-LOCAL (csrc/conv_local_body.h: whole samples x 16 filters x all of K per work item, filter
    slice resident in LDS before the item's flags arrive, no split-K seam, GroupNorm inside one wave)."""
    return os.environ.get("LFVDM_CHAIN_LOCAL", "1") != "0"


CHAIN_MAX_M = int(os.environ.get("LFVDM_CHAIN_MAX_M", "640"))
CHAIN_TIMEOUT_S = float(os.environ.get("LFVDM_CHAIN_TIMEOUT_S", "2.0"))


class _TableBudget:
    """Process-wide pool for the samplers' timestep tables (LFVDM_TIME_TABLE_GB, default 12: the headline chain needs
    6.8 GiB, a 288 GB card has room, but the pool is shared by ALL live sampler plans and released when a plan dies)."""

    def __init__(self):
        self._held = {}
        self._logged = set()

    def limit(self):
        return float(os.environ.get("LFVDM_TIME_TABLE_GB", "12")) * 2 ** 30

    def used(self):
        return sum(self._held.values())

    def remaining(self):
        return max(0.0, self.limit() - self.used())

    def reserve(self, plan, nbytes):
        if nbytes > self.remaining():
            import gc
            gc.collect()            # plans of dropped samplers may only be waiting for the cycle collector
        if nbytes > self.remaining():
            return False
        key = id(plan)
        self._held[key] = nbytes
        weakref.finalize(plan, self._held.pop, key, None)
        return True

    def log(self, msg):
        if os.environ.get("LFVDM_QUIET", "0") != "1" and msg not in self._logged:
            self._logged.add(msg)
            print("[lfvdm] " + msg, file=sys.stderr, flush=True)


_table_budget = _TableBudget()


class Plan:
    """Launch sequence + workspaces for one (B, T, H, W)."""

    def __init__(self, engine, B, T, H, W, want_attn, time_steps=None):
        # time_steps = n: the plan belongs to a sampler that walks a known schedule of n timesteps.  Everything the
        # network derives from (t, frame_indices) alone is then TABULATED once per chain (build_time_tables /
        # build_R_tables, same kernels on a virtual batch of n*B rows) and the per-step embedding / RPE launches leave
        # the denoising step: the step's clock kernel fetches the FiLM rows of its timestep, the temporal attention reads
        # its R slices straight from the tables (lfvdm_attn_temporal_sel).
        self.time_steps = int(time_steps) if time_steps else 0
        self.engine = engine
        self.model = engine.model
        self.B, self.T, self.H, self.W = B, T, H, W
        self.want_attn = want_attn
        self.dev = next(self.model.parameters()).device
        self.keep = []      # tensors / ctypes objects that must outlive the plan's raw pointers
        self.steps = []     # (callable, args) executed in order; every callable takes the stream last
        self.packs = []     # (conv weight parameter, packed buffer)
        self.attn_t, self.attn_s = [], []
        self.chains = []    # persistent level chains (build_chains): dicts with the device tables and the steps they replace
        self._build()

    # ------------------------------------------------------------------ helpers
    def buf(self, *shape, dtype=th.float32):
        t = th.empty(*shape, device=self.dev, dtype=dtype)
        self.keep.append(t)
        return t

    def scratch(self, key, rows, cols):
        """Shared scratch tensor [rows][cols] (launches are stream-ordered; a larger request replaces the buffer for
        later steps, earlier steps keep theirs)."""
        if level_chain_enabled() and rows <= CHAIN_MAX_M:
            # low-resolution levels: a persistent level chain has no launch boundary to order the reuse of a buffer, so
            # every use gets its own (a few hundred KB each)
            return self.buf(rows * cols).view(rows, cols)
        pool = self.__dict__.setdefault("_scratch", {})
        t = pool.get(key)
        if t is None or t.numel() < rows * cols:
            t = pool[key] = self.buf(rows * cols)
        return t[:rows * cols].view(rows, cols)

    def gn_apply(self, a, b, C0, C1, N, P, gn, film, act, key):
        out = self.scratch(key, N * P, C0 + C1)
        L = nat.lib()
        args = (_p(a), _p(b) if b is not None else None, C0, C1, N, P, _p(gn.weight), _p(gn.bias),
                _p(film) if film is not None else None, self.T if film is not None else 1,
                self.rows_ld if film is not None else 0, gn.eps, act, _p(out), None, None, None)
        need = int(L.lfvdm_gn_apply_ws_floats(C0 + C1, N, P))
        if need:        # large map: chunk statistics + apply over thousands of workgroups (see lfvdm_gn_apply_ws)
            ws = self.scratch("gn_ws", 1, need)
            self.add(L.lfvdm_gn_apply_ws, *args, _p(ws), need)
        else:
            self.add(L.lfvdm_gn_apply, *args)
        return out

    def packed(self, weight, c0=0, c1=None):
        """Packed [Cout][tap][Cin] copy of an OIHW conv weight, or of its input channels [c0, c1) (one half of a conv over a
        channel concat run as two convolutions)."""
        Cout, Cin, k, _ = weight.shape
        c1 = Cin if c1 is None else c1
        out = self.buf(Cout, k * k, c1 - c0)
        self.packs.append((weight, out) if (c0, c1) == (0, Cin) else (weight, out, c0, c1))
        return out

    def add(self, fn, *args):
        self.steps.append((fn, args))

    def add_conv(self, **kw):
        self.add_conv_args(self.conv_args(**kw))

    def add_conv_args(self, a):
        self.keep.append(a)
        self.add(nat.lib().lfvdm_conv_igemm, C.byref(a))

    def conv_fused_gn(self, **kw):
        """Add a conv whose epilogue also evaluates the NEXT GroupNorm(+FiLM)+activation into kw['gn_out'] - if a tile
        exists that holds whole samples and groups (low-resolution levels); returns False (nothing added) otherwise."""
        if not GN_EPILOGUE:
            return False
        a = self.conv_args(**kw)
        nt, nw = C.c_int(), C.c_int()
        if nat.lib().lfvdm_conv_igemm_config(C.byref(a), C.byref(nt), C.byref(nw)) != 0:
            return False
        self.add_conv_args(a)
        return True

    def conv_args(self, **kw):
        a = nat.ConvArgs()
        g = kw.get
        a.src0 = _p(kw["src0"]); a.src1 = _p(g("src1")) if g("src1") is not None else None
        a.C0 = kw["C0"]; a.C1 = g("C1", 0); a.N = kw["N"]; a.Hs = kw["Hs"]; a.Ws = kw["Ws"]
        a.up = g("up", 0); a.stride = g("stride", 1); a.ksize = g("ksize", 3); a.Ho = kw["Ho"]; a.Wo = kw["Wo"]
        a.coefA = _p(g("coefA")) if g("coefA") is not None else None
        a.coefB = _p(g("coefB")) if g("coefB") is not None else None
        a.act = g("act", nat.ACT_NONE)
        a.W = _p(kw["W"]); a.bias = _p(g("bias")) if g("bias") is not None else None; a.Cout = kw["Cout"]
        a.s2src0 = _p(g("s2src0")) if g("s2src0") is not None else None
        a.s2src1 = _p(g("s2src1")) if g("s2src1") is not None else None
        a.s2C0 = g("s2C0", 0); a.s2C1 = g("s2C1", 0)
        a.W2 = _p(g("W2")) if g("W2") is not None else None
        a.bias2 = _p(g("bias2")) if g("bias2") is not None else None
        a.res = _p(g("res")) if g("res") is not None else None
        a.ldr = g("ldr", kw["Cout"])
        a.resA = _p(g("resA")) if g("resA") is not None else None
        a.resB = _p(g("resB")) if g("resB") is not None else None
        a.out = _p(kw["out"]); a.ldo = kw["ldo"]; a.out_mode = g("out_mode", nat.OUT_ROWS)
        if g("gn_out") is not None:     # fused GroupNorm(+FiLM)(+activation) of the output (lfvdm_conv_args)
            gn, film = kw["gn"], g("gn_film")
            a.gn_gamma, a.gn_beta, a.gn_out = _p(gn.weight), _p(gn.bias), _p(kw["gn_out"])
            a.gn_film = _p(film) if film is not None else None
            a.gn_film_ld = self.rows_ld if film is not None else 0
            a.gn_film_div = self.T if film is not None else 1
            a.gn_act, a.gn_skip_raw, a.gn_eps = g("gn_act", nat.ACT_NONE), int(g("gn_skip_raw", 0)), gn.eps
            a.gn_gw, a.gn_ld = int(g("gn_gw", 0)), int(g("gn_ld", 0))
        # shared split-K workspace (launches are stream-ordered, so one buffer serves every conv of the plan)
        if getattr(self, "splitk_ws", None) is None:
            self.splitk_ws = self.buf(2 * 1024 * 1024)                      # 8 MiB of slabs
            self.splitk_cnt = self.buf(4096, dtype=th.int32).zero_()          # tickets: zero once, self-cleaning
        a.splitk_ws, a.splitk_cnt = _p(self.splitk_ws), _p(self.splitk_cnt)
        a.splitk_ws_floats, a.splitk_cnt_ints = self.splitk_ws.numel(), self.splitk_cnt.numel()
        return a

    # ------------------------------------------------------------------ build
    def _build(self):
        m, L = self.model, nat.lib()
        B, T, H, W = self.B, self.T, self.H, self.W
        N = B * T
        if T > 32:
            raise RuntimeError("native temporal attention supports at most 32 frames per window")
        if m.dims != 2:
            raise NotImplementedError("only dims=2 is on the native path (the reference scripts use 2)")
        if not m.use_scale_shift_norm:
            raise NotImplementedError("use_scale_shift_norm=False is not on the native path (default True)")
        ch = m.model_channels
        ted = 4 * ch
        Cx = m.in_channels - 1

        # ---- static inputs
        self.x_in = self.buf(B, T, Cx, H, W)
        self.x0_in = self.buf(B, T, Cx, H, W)
        self.obs = self.buf(N)
        self.mask = self.buf(B, T)
        self.fi = self.buf(B, T, dtype=th.int64)
        half = ch // 2
        self.Bpad = (B + 3) // 4 * 4
        self.tin = self.buf(self.Bpad + half)
        self.tin.zero_()
        self.tin[self.Bpad:] = timestep_freqs(ch).to(self.dev)

        # ---- embeddings: 3 row-dot launches for the whole network.  All FiLM rows (then all RPE time projections)
        # live in ONE [B][rows_ld] buffer, so that a sampler can fetch a timestep's FiLM rows with one copy.
        from .unet import ResBlock, FactorizedAttentionBlock
        e0, self.emb = self.buf(B, ted), self.buf(B, ted)
        te0, te2 = m.time_embed[0], m.time_embed[2]
        heads = []       # (key, linear, in_mode): FiLM projections first, RPE time projections after them
        for mod in m.modules():
            if isinstance(mod, ResBlock):
                heads.append((mod, mod.emb_layers[1], 1))
        self.film_floats = sum(lin.weight.shape[0] for _, lin, _ in heads)
        for mod in m.modules():
            if isinstance(mod, FactorizedAttentionBlock):
                ta = mod.temporal_attention
                for r in (ta.rpe_q, ta.rpe_k, ta.rpe_v):
                    heads.append((r, r.rpe_net.embed_diffusion_time, 0))
        self.rows_ld = sum(lin.weight.shape[0] for _, lin, _ in heads)
        assert self.film_floats % 4 == 0 and self.rows_ld % 4 == 0
        self.rows = self.buf(B, self.rows_ld)
        self.film, self.tproj, self.row_off = {}, {}, {}
        off = 0
        for key, lin, mode in heads:
            O = lin.weight.shape[0]
            (self.film if mode == 1 else self.tproj)[key] = self.rows[:, off:off + O]
            self.row_off[key] = off
            off += O
        self._heads = heads

        def emb_jobs(M, tin, ldin, e0_, emb_, rows_, se0_=None, semb_=None):
            """Device job tables of the three launches for M batch rows: sinusoid + Linear, SiLU + Linear, all heads.
            se0_ / semb_: silu(e0) / silu(emb) materialised by the caller (lfvdm_silu) - the launches then read them in
            in_mode 0 instead of evaluating the SiLU once per OUTPUT row (what makes M = thousands of rows affordable)."""
            j0 = [nat.RowdotJob(_p(te0.weight), _p(te0.bias), _p(tin), _p(e0_), ch, ted, M, ldin, ted, 2, 0, 0)]
            j1 = [nat.RowdotJob(_p(te2.weight), _p(te2.bias), _p(se0_ if se0_ is not None else e0_), _p(emb_), ted, ted, M, ted, ted,
                                0 if se0_ is not None else 1, 0, 0)]
            jg, row0 = [], 0
            for key, lin, mode in heads:
                O = lin.weight.shape[0]
                src, md = (semb_, 0) if (mode == 1 and semb_ is not None) else (emb_, mode)
                jg.append(nat.RowdotJob(_p(lin.weight), _p(lin.bias), _p(src), _p(rows_) + 4 * self.row_off[key], ted, O, M, ted,
                                        self.rows_ld, md, row0, 0))
                row0 += O
            return [nat.jobs_to_device(j, self.dev) for j in (j0, j1, jg)], len(jg), row0

        self._emb_jobs = emb_jobs

        def rpe_jobs(Bn, tproj_base, R_of):
            """Job table of the grouped RPE-network launch for Bn batch rows (R_of: rpe module -> output tensor)."""
            jobs, tile0 = [], 0
            tiles_per = (Bn * T * T + 31) // 32
            for mod in m.modules():
                if isinstance(mod, FactorizedAttentionBlock):
                    ta = mod.temporal_attention
                    for r in (ta.rpe_q, ta.rpe_k, ta.rpe_v):
                        net = r.rpe_net
                        jobs.append(nat.RpeJob(_p(tproj_base) + 4 * self.row_off[r], _p(net.embed_distances.weight),
                                               _p(net.embed_distances.bias), _p(net.out.weight), _p(net.out.bias), _p(R_of[r]),
                                               net.channels, tile0, self.rows_ld, 0, None))
                        tile0 += tiles_per
            return (nat.jobs_to_device(jobs, self.dev) if jobs else None), len(jobs), tile0

        self._rpe_jobs = rpe_jobs
        rpe_mods = [r for mod in m.modules() if isinstance(mod, FactorizedAttentionBlock)
                    for r in (mod.temporal_attention.rpe_q, mod.temporal_attention.rpe_k, mod.temporal_attention.rpe_v)]
        for r in rpe_mods:
            if r.rpe_net.channels > 512:
                raise RuntimeError("RPE nets support at most 512 channels natively")

        # every parameter the tables are computed from: their versions are part of the tables' signature (a torch-side
        # update limited to the embedding / RPE parameters - partial load_state_dict, p.data.copy_ - must rebuild them)
        self._time_params = [te0.weight, te0.bias, te2.weight, te2.bias] + [t for _, lin, _ in heads for t in (lin.weight, lin.bias)]
        for r in rpe_mods:
            net = r.rpe_net
            self._time_params += [net.embed_distances.weight, net.embed_distances.bias, net.out.weight, net.out.bias]

        # ---- timestep tables (sampler plans): decide now, the launch sequence differs.  The budget is ONE pool per
        # process shared by every live sampler plan (a long-video schedule keeps a sampler per window shape).
        self.t_sel = None
        self.time_table_bytes = 0
        self.time_table_fallback = None          # why the plan runs the per-step embedding / RPE launches instead
        # The R tables ([timestep][B][T][T][C] per RPE network: 6.3 GiB for the 1000 steps of the headline chain, linear in
        # batch and chain length) are kept as a ROLLING WINDOW of `time_ring` timesteps (LFVDM_TIME_RING, default 128; 0 =
        # the whole chain): timestep t lives in slot t % ring, the window is refilled half a ring at a time by the same
        # grouped RPE launch (ensure_R, called by the sampler between graph launches) - same total work per chain, same
        # values bit for bit, an eighth of the memory: batch 8 and the pixel-space plan fit the default budget.  The FiLM /
        # time-projection rows of all timesteps (rows_all, 80 MB) stay whole.
        self.time_ring = 0
        if self.time_steps:
            ring = int(os.environ.get("LFVDM_TIME_RING", "128")) // 16 * 16      # two halves, each a multiple of 8 steps
            if 0 < ring < self.time_steps:
                self.time_ring = ring
            Bv = self.time_steps * B
            table_bytes = 4 * B * (self.time_steps * self.rows_ld +
                                   (self.time_ring or self.time_steps) * T * T * sum(r.rpe_net.channels for r in rpe_mods))
            if os.environ.get("LFVDM_TIME_TABLES", "1") == "0":
                self.time_table_fallback = "disabled (LFVDM_TIME_TABLES=0)"
            elif B > 64:
                self.time_table_fallback = "batch > 64"
            elif not _table_budget.reserve(self, table_bytes):
                self.time_table_fallback = (f"{table_bytes / 2 ** 30:.1f} GiB of tables do not fit the remaining "
                                            f"{_table_budget.remaining() / 2 ** 30:.1f} GiB of LFVDM_TIME_TABLE_GB")
            if self.time_table_fallback:
                self.time_steps = 0
                _table_budget.log(f"timestep tables off for plan (B={B}, T={T}, {H}x{W}): {self.time_table_fallback}; "
                                  "+4 launches per step")
            else:
                self.time_table_bytes = table_bytes
                _table_budget.log(f"timestep tables on for plan (B={B}, T={T}, {H}x{W}): {table_bytes / 2 ** 30:.2f} GiB "
                                  f"({_table_budget.used() / 2 ** 30:.2f} of {_table_budget.limit() / 2 ** 30:.1f} GiB in use)")
        self.R = {}
        if self.time_steps:
            Bv = self.time_steps * B
            self.t_sel = self.buf(B, dtype=th.int64).zero_()          # the sampler's device-side step counter
            self.rows_all = self.buf(Bv, self.rows_ld)
            for r in rpe_mods:
                self.R[r] = self.buf((self.time_ring or self.time_steps) * B, T, T, r.rpe_net.channels)   # [slot][B][T][T][C]
            self.tables_sig = None
            self._fill_jobs = {}       # (t0, t1, slot0) -> job table of that refill (pointers only: reused by every chain)
            self._loaded = {}          # ring half -> block of timesteps it holds
            self._fi_rep = None
        else:
            (j0, j1, jg), n_g, rows_g = emb_jobs(B, self.tin, self.Bpad, e0, self.emb, self.rows)
            self.keep += [j0, j1, jg]
            self.add(L.lfvdm_rowdot, _p(j0), 1, ted)
            self.add(L.lfvdm_rowdot, _p(j1), 1, ted)
            self.add(L.lfvdm_rowdot, _p(jg), n_g, rows_g)
            # ---- all RPE networks in one launch
            for r in rpe_mods:
                self.R[r] = self.buf(B, T, T, r.rpe_net.channels)
            jr, n_r, tiles_r = rpe_jobs(B, self.rows, self.R)
            if jr is not None:
                self.keep.append(jr)
                self.add(L.lfvdm_rpe_nets_maxc, _p(jr), n_r, tiles_r, _p(self.fi), B, T, max(r.rpe_net.channels for r in rpe_mods))

        # ---- scratch shared by all blocks (forward-only plan)
        maxC = max(mod.channels for mod in m.modules() if isinstance(mod, FactorizedAttentionBlock))
        maxCin = max(mod.channels for mod in m.modules() if isinstance(mod, ResBlock))
        Mmax = N * H * W
        self.s_qkv = self.buf(Mmax * 3 * maxC)
        self.s_o = self.buf(Mmax * maxC)
        self.s_xn = self.buf(Mmax * maxC)

        # ---- network body
        conv0 = m.input_blocks[0][0]
        h0 = self.buf(N * H * W, ch)
        self.add(L.lfvdm_conv_in, _p(self.x_in), _p(self.x0_in), _p(self.obs), _p(conv0.weight), _p(conv0.bias), _p(h0),
                 N, Cx, H, W, ch)
        cur = dict(parts=[(h0, ch)], H=H, W=W)
        hs = [cur]
        stages = list(m.input_blocks)[1:] + [m.middle_block]
        outs = list(m.output_blocks)
        for i, blk in enumerate(stages):
            # the layer that consumes this stage's output (when it is not a concat): lets the stage's last GEMM
            # evaluate that layer's first GroupNorm + SiLU in its epilogue at the low-resolution levels
            nxt = stages[i + 1][0] if i + 1 < len(stages) else None
            if blk is m.middle_block:       # its output is concatenated with the last skip by output_blocks[0]
                cur = self._stage(blk, cur, after_cat=(outs[0][0], hs[-1]) if outs else None)
            else:
                cur = self._stage(blk, cur, after=nxt)
                hs.append(dict(parts=cur["parts"], H=cur["H"], W=cur["W"]))      # (the skip carries no pre-activation)
        for i, blk in enumerate(outs):
            skip = hs.pop()
            assert skip["H"] == cur["H"] and len(cur["parts"]) == 1 and len(skip["parts"]) == 1
            cat_act = cur.get("act1cat")            # GroupNorm + SiLU of the concat, already evaluated half by half
            nxt_cat = (outs[i + 1][0], hs[-1]) if i + 1 < len(outs) and hs else None
            last_blk = i + 1 == len(outs)
            cur = self._stage(blk, dict(parts=cur["parts"] + skip["parts"], H=cur["H"], W=cur["W"],
                                        **({"act1": cat_act} if cat_act is not None else {})), after_cat=nxt_cat,
                              after_gn=m.out[0] if last_blk and NEXT_GN_EPILOGUE else None)
            if not last_blk:
                cur.pop("act1", None)
        # head: GN + SiLU + 3x3 conv straight into the (B,T,C,H,W) layout
        (hb, hc), = cur["parts"]
        gn, conv = m.out[0], m.out[2]
        self.out = self.buf(B, T, m.out_channels, H, W)
        act = cur.get("act1")       # the last block's final projection may have evaluated the head's GroupNorm + SiLU already
        if act is None:
            act = self.gn_apply(hb, None, hc, 0, N, H * W, gn, None, nat.ACT_SILU, "act1")
        wp = self.packed(conv.weight)
        self.add_conv(src0=act, C0=hc, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=wp, bias=conv.bias,
                      Cout=m.out_channels, out=self.out, ldo=m.out_channels, out_mode=nat.OUT_NCHW)
        self.head = dict(act=act, Wp=wp, bias=conv.bias, C=hc, Cout=m.out_channels, step=len(self.steps) - 1)
        self.head_fused = False

    def _stage(self, blk, cur, after=None, after_cat=None, after_gn=None):
        """after: the layer that consumes the stage's output as it is; after_cat = (ResBlock, skip): the decoder ResBlock that
        consumes it CONCATENATED with a skip tensor (unet.py:460); after_gn: the GroupNorm (+ SiLU) the consumer applies first
        when the consumer is not a ResBlock (the U-Net head, unet.py:418-422)."""
        from .unet import ResBlock, FactorizedAttentionBlock, Downsample, Upsample
        layers = list(blk)
        for i, layer in enumerate(layers):
            last = i + 1 == len(layers)
            nxt = layers[i + 1] if not last else after
            ngn = nxt.in_layers[0] if (isinstance(nxt, ResBlock) and NEXT_GN_EPILOGUE) else None
            if last and after_gn is not None:
                ngn = after_gn
            self._cat_next = None
            if last and after_cat is not None and isinstance(after_cat[0], ResBlock) and NEXT_GN_EPILOGUE:
                self._cat_next = after_cat          # read by _final_conv of this layer's last GEMM
            if isinstance(layer, ResBlock):
                cur = self._res(layer, cur, ngn)
            elif isinstance(layer, FactorizedAttentionBlock):
                cur = self._attn(layer, cur, ngn)
            elif isinstance(layer, Downsample):
                cur = self._resample(layer.op, cur, True, ngn)
            elif isinstance(layer, Upsample):
                cur = self._resample(layer.conv, cur, False, None)
            else:
                raise NotImplementedError(type(layer))
        self._cat_next = None
        return cur

    def _final_conv(self, kw, next_gn, M):
        """Add a layer's last GEMM; if the consumer is a ResBlock, try to evaluate its first GroupNorm + SiLU in the
        epilogue (whole samples per tile: maps of <= 64 pixels) -> the pre-activated tensor, or None."""
        if next_gn is not None and next_gn.weight.shape[0] == kw["Cout"]:
            actn = self.scratch("actn", M, kw["Cout"])
            if self.conv_fused_gn(gn=next_gn, gn_out=actn, gn_act=nat.ACT_SILU, gn_skip_raw=0, **kw):
                return actn
            # a dense 1x1 projection with a residual on 16x16 frames (an attention block's output projection): projection,
            # residual and the consumer's GroupNorm + SiLU in one launch of (frame, 16 channels) workgroups
            P = kw["Ho"] * kw["Wo"]
            if (PROJ_GN and kw.get("ksize") == 1 and kw.get("res") is not None and kw.get("ldr") == kw["Cout"]
                    and kw.get("ldo") == kw["Cout"] and kw["C0"] == kw["Cout"] and not kw.get("src1") and not kw.get("s2src0")
                    and kw["Hs"] * kw["Ws"] == P and nat.lib().lfvdm_proj_gn_ok(kw["N"], P, kw["Cout"]) == 0):
                self.add(nat.lib().lfvdm_proj_gn, _p(kw["src0"]), _p(kw["W"]), _p(kw["bias"]), _p(kw["res"]), _p(next_gn.weight),
                         _p(next_gn.bias), next_gn.eps, nat.ACT_SILU, _p(actn), _p(kw["out"]), kw["N"], P, kw["Cout"])
                return actn
        cat = getattr(self, "_cat_next", None)
        if cat is not None and self._final_conv_cat(kw, cat, M):
            return None          # (the consumer finds the normalised concat in self._cat_done)
        self.add_conv(**kw)
        return None

    def _final_conv_cat(self, kw, cat, M):
        """The consumer is a decoder ResBlock that normalises concat(this output, skip) first (unet.py:460, :152-155).  With
        gw = (C0 + C1) / 32 dividing both halves no group straddles the concat, so GroupNorm32 + SiLU of the concat is the
        two halves normalised on their own: the left half in THIS GEMM's epilogue (gn_gw / gn_ld: written into the left
        columns of the consumer's operand), the skip half by one lfvdm_gn_apply_part that depends on nothing the decoder
        computes - in a persistent level chain it runs ahead, off the critical path, where lfvdm_gn_apply on the concat was
        a stage of its own between two GEMMs (2.3 us + two flag hops, four times per denoising step).  Low-resolution
        levels only (tiles that hold whole samples; M <= LFVDM_CHAIN_MAX_M).  -> True if the steps were added."""
        rb, skip = cat
        (sb, C1), = skip["parts"]
        C0, L = kw["Cout"], nat.lib()
        N, P = kw["N"], kw["Ho"] * kw["Wo"]
        Ccat = C0 + C1
        gw = Ccat // 32
        gn = rb.in_layers[0]
        if (not level_chain_enabled() or M > CHAIN_MAX_M or gn.weight.shape[0] != Ccat or Ccat % 32 or C0 % gw or C1 % gw
                or gw not in (2, 4, 8, 16) or C1 % 64 or P > 256 or skip["H"] * skip["W"] != P):
            return False
        if self._cat_split_ok(rb, C0, C1, N, kw["Ho"], kw["Wo"]):
            # SAMPLE-LOCAL chains: the consumer's first convolution over the concat (K = 9 (C0 + C1): its filter slice does not
            # fit the LDS next to the activations) runs as TWO convolutions - the skip half, which depends on nothing the
            # decoder computes, ahead of time into a partial tensor; the decoder half with that partial as its residual.  The
            # two halves of the normalised concat are then two dense operands.
            actL, actR = self.buf(M, C0), self.buf(M, C1)
            a = self.conv_args(gn=gn, gn_out=actL, gn_act=nat.ACT_SILU, gn_skip_raw=0, gn_gw=gw, gn_ld=C0, **kw)
            nt, nw = C.c_int(), C.c_int()
            if GN_EPILOGUE and L.lfvdm_conv_igemm_config(C.byref(a), C.byref(nt), C.byref(nw)) == 0:
                self.part_bases = getattr(self, "part_bases", {})
                self.part_bases[_p(actR)] = (_p(actR), 0)
                self.add(L.lfvdm_gn_apply_part, _p(sb), C1, N, P, gw, _p(gn.weight) + 4 * C0, _p(gn.bias) + 4 * C0, gn.eps,
                         nat.ACT_SILU, _p(actR), C1)
                self.skip_side = getattr(self, "skip_side", [])
                self.skip_side.append((self.steps[-1], _p(sb)))         # (step, the tensor it waits for)
                self.add_conv_args(a)
                self._cat_done = (actL, actR)
                return True
        act = self.buf(M, Ccat)
        a = self.conv_args(gn=gn, gn_out=act, gn_act=nat.ACT_SILU, gn_skip_raw=0, gn_gw=gw, gn_ld=Ccat, **kw)
        nt, nw = C.c_int(), C.c_int()
        if not GN_EPILOGUE or L.lfvdm_conv_igemm_config(C.byref(a), C.byref(nt), C.byref(nw)) != 0:
            return False
        out_ptr = _p(act) + 4 * C0
        self.part_bases = getattr(self, "part_bases", {})
        self.part_bases[out_ptr] = (_p(act), C0)
        self.add(L.lfvdm_gn_apply_part, _p(sb), C1, N, P, gw, _p(gn.weight) + 4 * C0, _p(gn.bias) + 4 * C0, gn.eps, nat.ACT_SILU,
                 out_ptr, Ccat)
        self.add_conv_args(a)
        self._cat_done = act
        return True

    def _cat_split_ok(self, rb, C0, C1, N, H, W):
        """Should the first convolution of decoder ResBlock ``rb`` over concat(h [C0], skip [C1]) run as two convolutions?
        Only where the sample-local chain stage takes the halves but not the whole (LDS)."""
        if not (chain_local_enabled() and level_chain_enabled()) or os.environ.get("LFVDM_CAT_SPLIT", "1") == "0":
            return False
        L = nat.lib()

        def ok(Cin):
            a = nat.ConvArgs()
            a.src0, a.W, a.out = 0x1000, 0x1000, 0x1000
            a.C0, a.N, a.Hs, a.Ws, a.Ho, a.Wo, a.stride, a.ksize, a.Cout, a.ldo, a.ldr = Cin, N, H, W, H, W, 1, 3, rb.out_channels, rb.out_channels, rb.out_channels
            a.out_mode = nat.OUT_ROWS
            return any(L.lfvdm_chain_local_ok(C.byref(a), rt) == 0 for rt in (1, 2))

        return N * H * W <= CHAIN_MAX_M and not ok(C0 + C1) and ok(C0) and ok(C1)

    def _res(self, rb, cur, next_gn=None):
        L = nat.lib()
        N, H, W = self.B * self.T, cur["H"], cur["W"]
        P = H * W
        parts = cur["parts"]
        (a, C0) = parts[0]
        (b, C1) = parts[1] if len(parts) > 1 else (None, 0)
        Cin, Cout = C0 + C1, rb.out_channels
        assert Cin == rb.channels
        if rb.use_conv:
            raise NotImplementedError("ResBlock(use_conv=True) skip is not used by the reference factory")
        gn1, conv1 = rb.in_layers[0], rb.in_layers[2]
        gn2, conv2 = rb.out_layers[0], rb.out_layers[3]
        pb = _p(b) if b is not None else None
        h1 = self.buf(N * P, Cout)
        film = self.film[rb]
        out = self.buf(N * P, Cout)
        # GroupNorm(+FiLM)+SiLU evaluated ONCE into a scratch tensor (the concat of the two sources becomes
        # real); the implicit GEMMs then stage raw operands (see lfvdm_gn_apply for why this wins on gfx950)
        act1 = cur.get("act1")          # already evaluated by the producer's epilogue?
        if act1 is None:
            act1 = self.gn_apply(a, b, C0, C1, N, P, gn1, None, nat.ACT_SILU, "act1")
        if isinstance(act1, tuple):
            # the two halves of the normalised concat as dense operands (_final_conv_cat, split form): the skip half first,
            # into a partial tensor that the decoder half takes as its residual
            actL, actR = act1
            part = self.buf(N * P, Cout)
            self.add_conv(src0=actR, C0=C1, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=self.packed(conv1.weight, C0, C0 + C1), Cout=Cout,
                          out=part, ldo=Cout)
            self.skip_side.append((self.steps[-1], _p(actR)))
            c1 = dict(src0=actL, C0=C0, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=self.packed(conv1.weight, 0, C0), bias=conv1.bias,
                      Cout=Cout, out=h1, ldo=Cout, res=part, ldr=Cout)
        else:
            c1 = dict(src0=act1, C0=Cin, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=self.packed(conv1.weight), bias=conv1.bias,
                      Cout=Cout, out=h1, ldo=Cout)
        # low-resolution levels: GroupNorm-2 + FiLM + SiLU in the epilogue of conv1 (whole samples per tile);
        # the raw h1 is not needed by anything else and is not written
        act2 = self.scratch("act2", N * P, Cout)
        if not self.conv_fused_gn(gn=gn2, gn_film=film, gn_out=act2, gn_act=nat.ACT_SILU, gn_skip_raw=1, **c1):
            self.add_conv(**c1)
            act2 = self.gn_apply(h1, None, Cout, 0, N, P, gn2, film, nat.ACT_SILU, "act2")
        kw = dict(src0=act2, C0=Cout, N=N, Hs=H, Ws=W, Ho=H, Wo=W, W=self.packed(conv2.weight), bias=conv2.bias,
                  Cout=Cout, out=out, ldo=Cout)
        if isinstance(rb.skip_connection, nn.Identity):
            assert b is None
            kw.update(res=a, ldr=Cout)
        else:
            sk = rb.skip_connection
            kw.update(s2src0=a, s2src1=b, s2C0=C0, s2C1=C1, W2=sk.weight, bias2=sk.bias)
        nact = self._final_conv(kw, next_gn, N * P)
        res = dict(parts=[(out, Cout)], H=H, W=W)
        if nact is not None:
            res["act1"] = nact
        if getattr(self, "_cat_done", None) is not None:
            res["act1cat"], self._cat_done = self._cat_done, None
        return res

    def _attn(self, ab, cur, next_gn=None):
        L = nat.lib()
        B, T = self.B, self.T
        N, H, W = B * T, cur["H"], cur["W"]
        P = H * W
        (x, Cc), = cur["parts"]
        M = N * P
        heads = ab.num_heads
        ta, sa = ab.temporal_attention, ab.spatial_attention
        # --- temporal: GN over (C/32 x T) per (b, pixel); residual on the normalised tensor (rpe.py:136,172)
        if TEMPORAL_QKV and L.lfvdm_gn_temporal_qkv_ok(B, T, P, Cc) == 0:
            # both are local to a (b, pixel) column of T rows: one launch (statistics in registers, qkv on MFMA from an LDS image)
            self.add(L.lfvdm_gn_temporal_qkv, _p(x), _p(ta.norm.weight), _p(ta.norm.bias), ta.norm.eps, _p(self.s_xn),
                     _p(ta.qkv.weight), _p(ta.qkv.bias), _p(self.s_qkv), B, T, P, Cc)
        else:
            self.add(L.lfvdm_gn_temporal, _p(x), _p(ta.norm.weight), _p(ta.norm.bias), ta.norm.eps, _p(self.s_xn), B, T, P, Cc)
            self.add_conv(src0=self.s_xn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=ta.qkv.weight, bias=ta.qkv.bias,
                          Cout=3 * Cc, out=self.s_qkv, ldo=3 * Cc)
        at = None
        if self.want_attn:
            at = self.buf(B * P, heads, T, T)
            self.attn_t.append(at)
        if self.time_steps:     # R tensors are tables over the chain's timesteps: slice t_sel[b]
            self.add(L.lfvdm_attn_temporal_ring, _p(self.s_qkv), _p(self.R[ta.rpe_q]), _p(self.R[ta.rpe_k]), _p(self.R[ta.rpe_v]),
                     _p(self.mask), _p(self.s_o), _p(at) if at is not None else None, B, T, P, Cc, heads, _p(self.t_sel),
                     self.time_ring)
        else:
            self.add(L.lfvdm_attn_temporal, _p(self.s_qkv), _p(self.R[ta.rpe_q]), _p(self.R[ta.rpe_k]), _p(self.R[ta.rpe_v]),
                     _p(self.mask), _p(self.s_o), _p(at) if at is not None else None, B, T, P, Cc, heads)
        yt = self.buf(M, Cc)
        proj = dict(src0=self.s_o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=ta.proj_out.weight,
                    bias=ta.proj_out.bias, Cout=Cc, res=self.s_xn, ldr=Cc, out=yt, ldo=Cc)
        # --- spatial: GN over (C/32 x HW) per frame.  Low-resolution levels: evaluated in the epilogue of the temporal
        # projection (whole frames per tile; the raw block output is read by nothing else)
        ysn = self.scratch("act1", M, Cc)
        if PROJ_GN and M > CHAIN_MAX_M and L.lfvdm_proj_gn_ok(N, P, Cc) == 0:
            # 16x16 (a frame is more rows than a GEMM tile) and 8x8 (the tile GEMM's GroupNorm epilogue: 8.5 us for 84 MFLOP):
            # GroupNorm units are independent - one launch of (frame, 16 channels) workgroups projects, adds the residual and
            # normalises (the raw sum is read by nothing else).  The chained levels keep the epilogue form (a chain stage).
            self.add(L.lfvdm_proj_gn, _p(self.s_o), _p(ta.proj_out.weight), _p(ta.proj_out.bias), _p(self.s_xn),
                     _p(sa.norm.weight), _p(sa.norm.bias), sa.norm.eps, nat.ACT_NONE, _p(ysn), None, N, P, Cc)
        elif not self.conv_fused_gn(gn=sa.norm, gn_out=ysn, gn_act=nat.ACT_NONE, gn_skip_raw=1, **proj):
            self.add_conv(**proj)
            ysn = self.gn_apply(yt, None, Cc, 0, N, P, sa.norm, None, nat.ACT_NONE, "act1")   # also the residual
        # qkv projection inside the attention launch where a frame, the head's filters and its q / k / v fit the LDS
        fused_sa = (SPATIAL_FUSED and not self.want_attn and L.lfvdm_attn_spatial_fused_ok(N, P, Cc, heads) == 0)
        if fused_sa:
            self.add(L.lfvdm_attn_spatial_fused, _p(ysn), _p(sa.qkv.weight), _p(sa.qkv.bias), _p(self.s_o), N, P, Cc, heads)
        else:
            self.add_conv(src0=ysn, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=sa.qkv.weight, bias=sa.qkv.bias,
                          Cout=3 * Cc, out=self.s_qkv, ldo=3 * Cc)
            asp = None
            if self.want_attn:
                asp = self.buf(N, heads, P, P)
                self.attn_s.append(asp)
            self.add(L.lfvdm_attn_spatial, _p(self.s_qkv), _p(self.s_o), _p(asp) if asp is not None else None, None, N, P, Cc, heads)
        ys = self.buf(M, Cc)
        nact = self._final_conv(dict(src0=self.s_o, C0=Cc, N=N, Hs=P, Ws=1, Ho=P, Wo=1, ksize=1, W=sa.proj_out.weight,
                                     bias=sa.proj_out.bias, Cout=Cc, res=ysn, ldr=Cc, out=ys, ldo=Cc), next_gn, M)
        res = dict(parts=[(ys, Cc)], H=H, W=W)
        if nact is not None:
            res["act1"] = nact
        if getattr(self, "_cat_done", None) is not None:
            res["act1cat"], self._cat_done = self._cat_done, None
        return res

    def _resample(self, conv, cur, down, next_gn=None):
        if not isinstance(conv, nn.Conv2d):
            raise NotImplementedError("conv_resample=False (pooling) is not on the native path")
        N, H, W = self.B * self.T, cur["H"], cur["W"]
        (x, Cc), = cur["parts"]
        Ho, Wo = (((H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1) if down else (2 * H, 2 * W))
        out = self.buf(N * Ho * Wo, Cc)
        nact = self._final_conv(dict(src0=x, C0=Cc, N=N, Hs=H, Ws=W, up=0 if down else 1, stride=2 if down else 1, Ho=Ho, Wo=Wo,
                                     W=self.packed(conv.weight), bias=conv.bias, Cout=Cc, out=out, ldo=Cc), next_gn, N * Ho * Wo)
        res = dict(parts=[(out, Cc)], H=Ho, W=Wo)
        if nact is not None:
            res["act1"] = nact
        if getattr(self, "_cat_done", None) is not None:
            res["act1cat"], self._cat_done = self._cat_done, None
        return res

    # ------------------------------------------------------------------ timestep tables (sampler plans)
    def build_time_tables(self, ts_table):
        """FiLM rows and RPE time projections of EVERY timestep of the chain: the three embedding launches of the
        per-step plan on a virtual batch of n_t*B rows, row (i, b) <- model timestep ts_table[i].  Bitwise the values the
        per-step launches produce (each output row depends on its own input row only)."""
        L, m = nat.lib(), self.model
        n_t, B = self.time_steps, self.B
        assert ts_table.numel() == n_t
        Bv, ch = n_t * B, m.model_channels
        ted, half = 4 * ch, ch // 2
        Bvp = (Bv + 3) // 4 * 4
        tin = th.zeros(Bvp + half, device=self.dev, dtype=th.float32)
        tin[:Bv] = ts_table.to(self.dev, th.float32).repeat_interleave(B)
        tin[Bvp:] = timestep_freqs(ch).to(self.dev)
        e0, emb = th.empty(Bv, ted, device=self.dev), th.empty(Bv, ted, device=self.dev)
        se0, semb = th.empty_like(e0), th.empty_like(emb)
        (j0, j1, jg), n_g, rows_g = self._emb_jobs(Bv, tin, Bvp, e0, emb, self.rows_all, se0, semb)
        s = nat.stream()
        nat.check(L.lfvdm_rowdot(_p(j0), 1, ted, s), "lfvdm_rowdot")
        nat.check(L.lfvdm_silu(_p(e0), _p(se0), e0.numel(), s), "lfvdm_silu")
        nat.check(L.lfvdm_rowdot(_p(j1), 1, ted, s), "lfvdm_rowdot")
        nat.check(L.lfvdm_silu(_p(emb), _p(semb), emb.numel(), s), "lfvdm_silu")
        nat.check(L.lfvdm_rowdot(_p(jg), n_g, rows_g, s), "lfvdm_rowdot")
        th.cuda.current_stream().synchronize()      # the temporaries and job tables above die with this frame
        self.tables_sig = (self.time_signature(), tuple(ts_table.tolist()))       # (compared by GraphSampler.begin)

    def build_R_tables(self, frame_indices):
        """Per chain: R_q / R_k / R_v of every temporal attention depend on the timestep and on THIS window's frame indices.
        Whole-chain tables (time_ring == 0): one grouped RPE launch on the virtual batch of n_t * B rows.  Rolling window:
        only remember the indices; ``ensure_R`` fills the halves of the ring as the chain reaches them."""
        if not self.R:
            return
        n_t, B, T = self.time_steps, self.B, self.T
        steps = (self.time_ring // 2) if self.time_ring else n_t
        self._fi_rep = frame_indices.to(self.dev, th.int64).reshape(B, T).repeat(steps, 1).contiguous()
        self._loaded = {}
        if not self.time_ring:
            self._fill_R(0, n_t, 0)
        # no host synchronisation: the index tensor is released to the caching allocator in stream order, and a windowed
        # sampler (97 chains per video) keeps the host ahead of the device across windows

    def _fill_R(self, t0, t1, slot0):
        """R rows of the timesteps [t0, t1) -> slots [slot0, slot0 + t1 - t0): the grouped RPE launch of the per-step plan
        on the virtual batch of (t1 - t0) * B rows (bitwise the per-step values: every output row depends on its own
        input row only)."""
        B, T = self.B, self.T
        key = (t0, t1, slot0)
        ent = self._fill_jobs.get(key)
        if ent is None:
            views = {r: buf[slot0 * B:] for r, buf in self.R.items()}
            ent = self._rpe_jobs((t1 - t0) * B, self.rows_all[t0 * B:], views)
            self._fill_jobs[key] = ent
        jr, n_r, tiles_r = ent
        nat.check(nat.lib().lfvdm_rpe_nets_maxc(_p(jr), n_r, tiles_r, _p(self._fi_rep), (t1 - t0) * B, T,
                                                max(r.rpe_net.channels for r in self.R), nat.stream()), "lfvdm_rpe_nets")

    def ensure_R(self, t_hi, t_lo=None):
        """Rolling window: make the R rows of the timesteps t_lo..t_hi resident (at most half a ring apart: the sampler
        calls this before every graph launch with the timesteps that launch walks).  Half h of the ring holds the block of
        ring / 2 timesteps blk with blk % 2 == h; a refill is one launch, stream-ordered behind the steps that read the
        half's previous content."""
        if not self.time_ring or not self.R:
            return
        H = self.time_ring // 2
        t_hi = min(max(int(t_hi), 0), self.time_steps - 1)
        t_lo = t_hi if t_lo is None else min(max(int(t_lo), 0), t_hi)
        assert t_hi - t_lo < H, "a graph launch walks less than half a ring of timesteps"
        for blk in {t_hi // H, t_lo // H}:
            if self._loaded.get(blk % 2) != blk:
                self._fill_R(blk * H, min((blk + 1) * H, self.time_steps), (blk % 2) * H)
                self._loaded[blk % 2] = blk

    def fill_whole_chain(self):
        """Every R block of a chain once (what a chain costs in table building; bench.py charges it per step)."""
        if not self.R:
            return
        if not self.time_ring:
            self._fill_R(0, self.time_steps, 0)
            return
        H = self.time_ring // 2
        for blk in range((self.time_steps + H - 1) // H):
            self._fill_R(blk * H, min((blk + 1) * H, self.time_steps), (blk % 2) * H)
        self._loaded = {}

    def fuse_head_update(self, t_buf, tables, clip, seed, noise, pred, inject_noise):
        """Sampler only (its plan is private): replace the last launch - the output convolution - by
        lfvdm_conv_out_psample, which also does the x_{t-1} update on ``x_in`` (reference gaussian_diffusion.py:369-401)
        with the chain's in-kernel noise (or ``noise`` as given when ``inject_noise``).  -> False if the shape is not
        covered (the sampler then issues the update as its own launch)."""
        L = nat.lib()
        h = self.head
        if self.head_fused:
            return True
        if h["step"] != len(self.steps) - 1 or L.lfvdm_conv_out_psample_ok(self.B * self.T, self.H, self.W, h["C"], h["Cout"]) != 0:
            return False
        self.keep.append((t_buf, tables, seed, noise, pred))
        args = (_p(h["act"]), _p(h["Wp"]), _p(h["bias"]), _p(self.out), _p(self.x_in), _p(noise) if inject_noise else None,
                None if inject_noise else _p(noise), _p(t_buf), _p(tables["sqrt_recip_alphas_cumprod"]),
                _p(tables["sqrt_recipm1_alphas_cumprod"]), _p(tables["posterior_mean_coef1"]),
                _p(tables["posterior_mean_coef2"]), _p(tables["model_log_variance"]), int(bool(clip)), _p(self.x_in), _p(pred),
                None, self.B, self.T, self.H, self.W, h["C"], h["Cout"], _p(seed))
        self.steps[h["step"]] = (L.lfvdm_conv_out_psample, args)
        self.head_fused = True
        return True

    def tick(self, t_buf, ts_table):
        """The sampler's clock (t <- max(t-1, 0); model timestep <- table[t]); with timestep tables it also fetches
        the FiLM rows of the new t."""
        L = nat.lib()
        if self.time_steps:
            nat.check(L.lfvdm_sampler_tick_fetch(_p(t_buf), _p(ts_table), _p(self.tin), self.B, _p(self.rows_all), self.rows_ld,
                                                 _p(self.rows), self.film_floats, nat.stream()), "lfvdm_sampler_tick_fetch")
        else:
            nat.check(L.lfvdm_sampler_tick(_p(t_buf), _p(ts_table), _p(self.tin), self.B, nat.stream()), "lfvdm_sampler_tick")

    # ------------------------------------------------------------------ autotune
    def autotune(self, rounds=3, reps=6):
        """Pin the fastest (tile shape, K-chunk, split-K) variant of each implicit-GEMM launch of this plan
        (``_native.tuned_code``: measured once per launch shape, shared by every plan and by the training path;
        the built-in makespan model is only the starting point)."""
        L = nat.lib()
        cache = nat.tune_cache()
        tuned = 0
        for fn, args in self.steps:
            if fn is not L.lfvdm_conv_igemm:
                continue
            a = args[0]._obj
            key = nat.tune_key(a)
            if (level_chain_enabled() and a.N * a.Ho * a.Wo <= CHAIN_MAX_M and a.out_mode == nat.OUT_ROWS
                    and self._local_tiles(a) == 0):
                # candidate stage of a persistent level chain: the fastest code among the variants the chain kernel holds
                # (own cache entry: the unrestricted choice of the same shape stays what the other callers get)
                ckey = key + (nat.TUNE_CHAIN,)
                if ckey not in cache:
                    cache[ckey] = nat.autotune_launch(a, rounds, reps, chain_only=True)
                if cache[ckey]:
                    a.tune = cache[ckey]
                    tuned += 1
                    continue
            if key not in cache:
                cache[key] = nat.autotune_launch(a, rounds, reps)
            a.tune = cache[key]
            tuned += 1
        self.tuned = True
        nat.tune_cache_save()
        n_chains = self.build_chains()
        # low-resolution launches that did not end up in a chain (a run of one, a refused plan) were given the fastest code
        # among the variants the CHAIN kernel holds: they run per launch, so they get the unrestricted choice back
        for fn, args in self.steps:
            if fn is L.lfvdm_conv_igemm:
                a = args[0]._obj
                key = nat.tune_key(a)
                if cache.get(key + (nat.TUNE_CHAIN,)) and a.tune == cache[key + (nat.TUNE_CHAIN,)]:
                    if key not in cache:
                        cache[key] = nat.autotune_launch(a, rounds, reps)
                    a.tune = cache[key]
        nat.tune_cache_save()
        if n_chains and os.environ.get("LFVDM_CHAIN_TUNE", "1") != "0":
            self.tune_chains()
        return tuned

    # ------------------------------------------------------------------ persistent level chains
    def _local_tiles(self, a):
        """Row tiles per work item (1 | 2) if this launch runs as a SAMPLE-LOCAL chain stage, else 0.  One tile of 16 rows
        per item while the stage then fits one round of 256 workgroups, else two."""
        L = nat.lib()
        if not (chain_local_enabled() and level_chain_enabled()) or a.N * a.Ho * a.Wo > CHAIN_MAX_M or a.out_mode != nat.OUT_ROWS:
            return 0
        forced = int(os.environ.get("LFVDM_CHAIN_LOCAL_RT", "0")) or int(nat.tune_cache().get(nat.tune_key(a) + (nat.TUNE_LOCAL_RT,), 0))
        M, NS = a.N * a.Ho * a.Wo, a.Cout // 16
        order = (1, 2) if -(-M // 16) * NS <= 256 else (2, 1)
        for rt in ((forced,) if forced else order):
            if L.lfvdm_chain_local_ok(C.byref(a), rt) == 0:
                return rt
        return 0

    def _chain_stage(self, step):
        """-> ChainStage for a step that can run inside a persistent level chain, else None."""
        L = nat.lib()
        fn, args = step
        st = nat.ChainStage()
        if fn is L.lfvdm_conv_igemm:
            a = args[0]._obj
            rt = self._local_tiles(a)
            if rt:
                st.kind, st.cfg = nat.CHAIN_LOCAL, rt
                C.memmove(C.byref(st.conv), C.byref(a), C.sizeof(nat.ConvArgs))
                return st
            if a.N * a.Ho * a.Wo > CHAIN_MAX_M or L.lfvdm_chain_conv_ok(C.byref(a)) != 0:
                return None
            st.kind = nat.CHAIN_CONV
            C.memmove(C.byref(st.conv), C.byref(a), C.sizeof(nat.ConvArgs))
            return st
        if fn is L.lfvdm_gn_apply_part:
            (src, Cp, N, P, cg, gamma, beta, eps, act, out, ldo) = args
            if N * P > CHAIN_MAX_M or L.lfvdm_chain_gn_ok(Cp, 0, N, P) != 0:
                return None
            st.kind = nat.CHAIN_GN
            g = st.gn
            g.src0, g.src1, g.C0, g.C1, g.N, g.P = src, None, Cp, 0, N, P
            g.gamma, g.beta, g.film, g.film_div, g.film_ld, g.eps, g.act, g.out = gamma, beta, None, 1, 0, eps, act, out
            g.cg, g.ldo = cg, ldo
            g.out_base, g.out_col = self.part_bases[out]
            return st
        if fn is L.lfvdm_gn_apply:
            (src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, act, out, cA, cB, stats) = args
            if N * P > CHAIN_MAX_M or cA or cB or stats or L.lfvdm_chain_gn_ok(C0, C1, N, P) != 0:
                return None
            st.kind = nat.CHAIN_GN
            g = st.gn
            g.src0, g.src1, g.C0, g.C1, g.N, g.P = src0, src1, C0, C1, N, P
            g.gamma, g.beta, g.film, g.film_div, g.film_ld, g.eps, g.act, g.out = gamma, beta, film, film_div, film_ld, eps, act, out
            return st
        return None

    def _make_chain(self, run, keep=True):
        """One lfvdm_level_chain step for a run of chainable steps (list of (step, ChainStage)), or None.  keep=False: the
        device tables live only as long as the returned dict (the in-chain tuner builds hundreds of candidates)."""
        L = nat.lib()
        n = len(run)
        # stages that read nothing the chain produces (the skip half of a concat GroupNorm) first: the planner puts them on
        # idle workgroups, and a workgroup walks the stages in list order - at the front they run at once
        def outs_of(st):
            return ({st.conv.out, st.conv.gn_out} if st.kind != nat.CHAIN_GN else {st.gn.out_base or st.gn.out}) - {None}

        def srcs_of(st):
            if st.kind == nat.CHAIN_GN:
                return {st.gn.src0, st.gn.src1} - {None}
            cv = st.conv
            return {cv.src0, cv.src1, cv.s2src0, cv.s2src1, cv.res} - {None}

        # free-standing stages: GroupNorms of tensors from earlier launches (the skip half of a concat normalisation) and
        # sample-local convolutions that read nothing else (the skip half of a split concat convolution)
        produced = set()
        for _, st in run:
            produced |= outs_of(st)
        free_gn = [st for _, st in run if st.kind == nat.CHAIN_GN and not (srcs_of(st) & produced)]
        gn_outs = set()
        for st in free_gn:
            gn_outs |= outs_of(st)
        free_ids = {id(st) for st in free_gn}
        free_ids |= {id(st) for _, st in run if st.kind == nat.CHAIN_LOCAL and srcs_of(st) and srcs_of(st) <= gn_outs}
        free_ids |= {id(st) for _, st in run if st.side}
        run = [e for e in run if id(e[1]) in free_ids] + [e for e in run if id(e[1]) not in free_ids]
        stages = (nat.ChainStage * n)(*[st for _, st in run])
        cap = 1 << 20
        deps = (C.c_int32 * cap)()
        used, ws_f, cnt_i = C.c_int64(), C.c_int64(), C.c_int64()
        n_flags, grid, lds = C.c_int32(), C.c_int32(), C.c_int32()
        rc = L.lfvdm_chain_plan(stages, n, deps, cap, C.byref(used), C.byref(n_flags), C.byref(ws_f), C.byref(cnt_i),
                                C.byref(grid), C.byref(lds))
        if rc != 0:
            return None
        # the chain's waits only end if ALL its workgroups are resident: ask the device how many it holds with this much
        # LDS (a partitioned / CU-masked / smaller part than the 256 CUs of a whole MI355X) and plan again under that cap
        if self.dev.type == "cuda":
            room = int(L.lfvdm_chain_capacity(lds.value))
            if room < max(8, min(64, grid.value)):
                _table_budget.log(f"persistent level chain refused: the device holds {room} of its workgroups at once")
                return None
            if grid.value > room:
                grid = C.c_int32(room // 8 * 8)
                rc = L.lfvdm_chain_plan(stages, n, deps, cap, C.byref(used), C.byref(n_flags), C.byref(ws_f), C.byref(cnt_i),
                                        C.byref(grid), C.byref(lds))
                if rc != 0 or grid.value > room:
                    return None
        ws = th.empty(max(1, ws_f.value), device=self.dev)
        cnt = th.zeros(max(1, cnt_i.value), dtype=th.int32, device=self.dev)
        for i in range(n):
            if stages[i].kind != nat.CHAIN_GN:
                cv = stages[i].conv
                cv.splitk_ws, cv.splitk_cnt = _p(ws) + 4 * stages[i].ws_off, _p(cnt) + 4 * stages[i].cnt_off
                cv.splitk_ws_floats, cv.splitk_cnt_ints = ws.numel() - stages[i].ws_off, cnt.numel() - stages[i].cnt_off
        stages_dev = th.frombuffer(bytearray(bytes(memoryview(stages))), dtype=th.uint8).to(self.dev)
        deps_dev = th.frombuffer(bytearray(bytes(memoryview(deps))[:4 * max(1, used.value)]), dtype=th.int32).to(self.dev)
        flags = th.zeros(max(1, n_flags.value), dtype=th.int32, device=self.dev)
        ctl = th.zeros(nat.CHAIN_CTL_INTS, dtype=th.int32, device=self.dev)
        ch = dict(steps=[st for st, _ in self._launch_order(run)], run=list(run), n=n, ctl=ctl, grid=grid.value, lds=lds.value,
                  items=[s.n_items for s in stages], kinds=[s.kind for s in stages], codes=[s.conv.tune for s in stages],
                  room=(room if self.dev.type == "cuda" else None),
                  tensors=[ws, cnt, stages_dev, deps_dev, flags, ctl])
        ch["step"] = (L.lfvdm_level_chain, (_p(stages_dev), n, _p(deps_dev), _p(flags), _p(ctl), grid.value, lds.value,
                                            CHAIN_TIMEOUT_S))
        if keep:
            self.keep += ch["tensors"]
        return ch

    def _launch_order(self, run):
        """The run in the order of the plan's step list (what ``disable_chains`` restores): hoisted stages go back to where
        their launches stood."""
        pos = self._step_pos          # (positions in the per-launch step list, recorded by build_chains)
        return sorted(run, key=lambda e: pos[id(e[0])])

    def _time_chain(self, ch, reps, rounds):
        """Microseconds per launch of one chain on its own, back to back (minimum over `rounds`)."""
        fn, args = ch["step"]
        s = nat.stream()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn(*args, s)
        best = float("inf")
        for _ in range(rounds):
            e0.record()
            for _ in range(reps):
                fn(*args, s)
            e1.record()
            e1.synchronize()
            best = min(best, 1000.0 * e0.elapsed_time(e1) / reps)
        if int(ch["ctl"][nat.CHAIN_CTL_ABORT].item()) != 0:
            return float("inf")
        return best

    def tune_chains(self, reps=20, rounds=3):
        """Tile codes of the chains' GEMM stages chosen INSIDE the chain (one pass of coordinate descent: every legal code
        of a stage with the other stages at their current choice, the chain timed on its own).  A stand-alone launch and a
        chain stage have different optima: in a chain the filter pieces are in flight before the wait, so a third LDS-DMA
        stage (the whole K slice of a k-group requested up front) pays where it did not per launch, and a stage with more
        work items than workgroups runs in rounds.  Cached per launch shape (key + TUNE_IN_CHAIN)."""
        if not self.chains or th.cuda.is_current_stream_capturing():
            return 0
        L, cache, changed = nat.lib(), nat.tune_cache(), 0
        reps = int(os.environ.get("LFVDM_TUNE_REPS", reps))
        rounds = int(os.environ.get("LFVDM_TUNE_ROUNDS", rounds))
        codes = (C.c_int * 256)()
        for ci in range(len(self.chains)):
            ch = self.chains[ci]
            run, cur, cur_t = ch["run"], ch, None
            for si, (step, st) in enumerate(run):
                if st.kind == nat.CHAIN_LOCAL:
                    # row tiles per item (1 | 2) of a sample-local stage, measured in the chain: one tile = more items (a second
                    # round where they outnumber the workgroups), two = half the filter fetches per output row
                    a = step[1][0]._obj
                    key = nat.tune_key(a) + (nat.TUNE_LOCAL_RT,)
                    if os.environ.get("LFVDM_CHAIN_LOCAL_RT"):
                        continue
                    if key in cache:                   # (measured on an earlier stage of the same shape, or in another plan)
                        if cache[key] != st.cfg and L.lfvdm_chain_local_ok(C.byref(st.conv), cache[key]) == 0:
                            old_rt, st.cfg = st.cfg, cache[key]
                            nxt = self._make_chain(run, keep=False)
                            if nxt is not None:
                                cur, cur_t = nxt, None
                            else:
                                st.cfg = old_rt
                        continue
                    other = 3 - st.cfg
                    best_rt = st.cfg
                    if L.lfvdm_chain_local_ok(C.byref(st.conv), other) == 0:
                        if cur_t is None:
                            cur_t = self._time_chain(cur, reps, rounds)
                        st.cfg = other
                        cand = self._make_chain(run, keep=False)
                        t = self._time_chain(cand, reps, rounds) if cand is not None else float("inf")
                        if t < cur_t * 0.995:
                            cur, cur_t, best_rt = cand, t, other
                        st.cfg = best_rt
                    cache[key] = best_rt
                    continue
                if st.kind != nat.CHAIN_CONV:
                    continue
                a = step[1][0]._obj
                key = nat.tune_key(a) + (nat.TUNE_IN_CHAIN,)
                if key in cache:
                    if cache[key] != st.conv.tune:
                        old_code = st.conv.tune
                        st.conv.tune = a.tune = cache[key]
                        nxt = self._make_chain(run, keep=False)
                        if nxt is not None:
                            cur, cur_t = nxt, None
                        else:       # the cached code does not plan in THIS chain: the stage keeps the code it runs with
                            st.conv.tune = a.tune = old_code
                    continue
                if cur_t is None:
                    cur_t = self._time_chain(cur, reps, rounds)
                n = L.lfvdm_conv_igemm_candidates(C.byref(a), codes, 256)
                start = st.conv.tune
                best_code = start
                for code in [codes[i] for i in range(n)]:
                    if code == start:
                        continue
                    st.conv.tune = code
                    if L.lfvdm_chain_conv_ok(C.byref(st.conv)) != 0:
                        continue
                    cand = self._make_chain(run, keep=False)
                    if cand is None:
                        continue
                    t = self._time_chain(cand, reps, rounds)
                    if t < cur_t * 0.995:
                        cur, cur_t, best_code = cand, t, code
                st.conv.tune = a.tune = best_code
                cache[key] = best_code
            if cur is not ch:
                changed += 1
                self.keep += cur["tensors"]
                idx = next(i for i, stp in enumerate(self.steps) if stp is ch["step"])
                self.steps[idx] = cur["step"]
                self.chains[ci] = cur
        nat.tune_cache_save()
        return changed

    def build_chains(self):
        """Replace every run of >= 2 consecutive chainable launches (tuned implicit GEMMs and one-wave GroupNorms of the
        low-resolution levels) by ONE persistent launch (lfvdm_level_chain).  Idempotent; called by ``autotune``."""
        if not level_chain_enabled() or self.chains or getattr(self, "chains_off", False):
            return 0
        head_last = self.head["step"] == len(self.steps) - 1
        self._step_pos = {id(sp): i for i, sp in enumerate(self.steps)}
        # runs of consecutive chainable steps
        runs, lone = [], []           # runs: [(first position, [(step, stage)...])]; lone: (position, step)
        run = []
        for i, step in enumerate(self.steps):
            st = self._chain_stage(step)
            if st is not None:
                run.append((step, st))
            else:
                if run:
                    runs.append((self._step_pos[id(run[0][0])], run))
                    run = []
                lone.append((i, step))
        if run:
            runs.append((self._step_pos[id(run[0][0])], run))
        # The SKIP SIDE of the decoder's split concat convolutions (GroupNorm of the encoder's skip tensor, then the skip half
        # of the convolution: _final_conv_cat / _res) depends on nothing the decoder computes: SIDE stages of their chain
        # (lfvdm_chain_stage.side: the planner reserves part of the grid for them), listed first and in the order in which the
        # main path needs them, they run beside it from the first microsecond - their operands come from earlier launches.
        # Measured alternatives: on shared workgroups at the front of the chain they kept the main path's first workgroups
        # busy for ~20 us; moved into the encoder-side chain (tail, shared workgroups) they started when that chain's main
        # path had ended (+34 us), and on a reserved quarter of that chain's grid they took 100 us for its 64.
        if os.environ.get("LFVDM_SKIP_SIDE", "1") != "0":
            for step, _ in getattr(self, "skip_side", []):
                for _, r in runs:
                    for sp, st in r:
                        if sp is step and any(not x.side and x is not st for _, x in r):
                            st.side = 1
        out = []
        for pos, step in sorted([(p, ("run", r)) for p, r in runs if r] + [(p, ("step", sp)) for p, sp in lone], key=lambda e: e[0]):
            if step[0] == "step":
                out.append(step[1])
                continue
            r = step[1]
            # (LFVDM_CHAIN_SINGLE=1: a lone sample-local launch - the 1x1 projections between the attention kernels of the middle
            # block - as a one-stage "chain".  Measured and left off: 7.7 us per launch against 5.6 for the tile kernel on
            # those 160-row, K = 128 GEMMs - with nothing to wait for, the filter fetch in front of the K loop is exposed)
            single = len(r) == 1 and r[0][1].kind == nat.CHAIN_LOCAL and os.environ.get("LFVDM_CHAIN_SINGLE", "0") == "1"
            ch = self._make_chain(r) if len(r) >= 2 or single else None
            if ch is None:
                out.extend(sp for sp, _ in r)
            else:
                self.chains.append(ch)
                out.append(ch["step"])
        self.steps = out
        if head_last:
            self.head["step"] = len(self.steps) - 1
        return len(self.chains)

    def disable_chains(self):
        """Back to one launch per stage (after a chain reported a timeout, or for A/B runs)."""
        if not self.chains:
            self.chains_off = True
            return
        head_last = self.head["step"] == len(self.steps) - 1
        by_step = {id(ch["step"]): ch for ch in self.chains}
        out = []
        for step in self.steps:
            ch = by_step.get(id(step))
            out.extend(ch["steps"] if ch is not None else [step])
        self.steps, self.chains, self.chains_off = out, [], True
        if head_last:
            self.head["step"] = len(self.steps) - 1

    def split_chains(self):
        """Every stage of every chain as a chain of its own (one launch per stage, the SAME kernel bodies and work items: no
        flag is waited for, the launch boundary orders the stages).  Bitwise the chained plan - the guard that a hand-off
        inside a chain never delivers a stale byte; also what a plan falls back to where it must not depend on co-residency
        but wants the same numerics."""
        if not self.chains:
            return 0
        head_last = self.head["step"] == len(self.steps) - 1
        by_step = {id(ch["step"]): ch for ch in self.chains}
        out, n = [], 0
        for step in self.steps:
            ch = by_step.get(id(step))
            if ch is None:
                out.append(step)
                continue
            for sp, st in self._launch_order(ch["run"]):
                one = self._make_chain([(sp, st)])
                assert one is not None
                out.append(one["step"])
                n += 1
        self.steps, self.chains, self.chains_off = out, [], True
        if head_last:
            self.head["step"] = len(self.steps) - 1
        return n

    def chains_aborted(self):
        """Did a wait inside a persistent level chain time out?  (synchronises: ONE read-back for all chains)"""
        if not self.chains:
            return False
        return bool(th.stack([ch["ctl"][nat.CHAIN_CTL_ABORT] for ch in self.chains]).ne(0).any().item())

    # ------------------------------------------------------------------ run
    def weight_signature(self):
        # parameter versions catch torch-side in-place updates (optimizers, load_state_dict); the engine
        # epoch is bumped explicitly by code that rewrites parameters through raw pointers (fused AdamW)
        return (self.engine.epoch,) + tuple(e[0]._version for e in self.packs)

    def time_signature(self):
        """Signature of what the timestep tables were computed from: the conv-weight signature plus the versions of the
        time-embedding MLP, the FiLM projections and the RPE networks."""
        return self.weight_signature() + tuple(t._version for t in self._time_params)

    def refresh_weights(self):
        """(Re)pack the OIHW conv weights into the [Cout][tap][Cin] layout the kernels read."""
        s = nat.stream()
        L = nat.lib()
        for w, out, *part in self.packs:
            if part:        # input channels [c0, c1): a contiguous OIHW copy first (released in stream order)
                w = w.detach()[:, part[0]:part[1]].contiguous()
            nat.check(L.lfvdm_pack_conv_weight(_p(w), _p(out), w.shape[0], w.shape[1], w.shape[2], s), "pack")
        self._sig = self.weight_signature()

    def launch(self, tick=None):
        """Enqueue the whole forward on the current stream (graph-capturable: no sync, no alloc).

        ``tick`` = (t_buf, ts_table), sampler only, timestep tables only: the clock (``Plan.tick``) rides in the first
        launch of the forward, lfvdm_conv_in_tick - nothing in the first conv reads the timestep.  (Forking the
        timestep-only launches onto a second graph branch was measured slower - DESIGN.md §5 - and is gone.)"""
        L = nat.lib()
        s = nat.stream()
        for fn, args in self.steps:
            if tick is not None and fn is L.lfvdm_conv_in:
                rc = L.lfvdm_conv_in_tick(*args, _p(tick[0]), _p(tick[1]), _p(self.tin), self.B, _p(self.rows_all),
                                          self.rows_ld, _p(self.rows), self.film_floats, s)
                tick = None
            else:
                rc = fn(*args, s)
            if rc:
                nat.check(rc, getattr(fn, "__name__", "kernel"))
        assert tick is None, "the plan has no lfvdm_conv_in launch to carry the clock"

    def set_inputs(self, x, x0, timesteps, frame_indices, obs_mask, latent_mask):
        B, T = self.B, self.T
        if x.data_ptr() != self.x_in.data_ptr():
            self.x_in.copy_(x)
        if x0.data_ptr() != self.x0_in.data_ptr():
            self.x0_in.copy_(x0)
        ob = obs_mask.reshape(B, T).to(th.float32)
        self.obs.copy_(ob.reshape(-1))
        th.clamp(ob + latent_mask.reshape(B, T).to(th.float32), max=1.0, out=self.mask)
        self.fi.copy_(frame_indices)
        self.tin[:B].copy_(timesteps)  # int or float timesteps -> float (reference nn.py:119)


class Engine:
    """Per-model cache of plans keyed by input shape."""

    def __init__(self, model):
        nat.lib()  # fail loudly if the native library is missing
        self.model = model
        self.plans = {}
        self.epoch = 0
        for p in model.parameters():
            if p.dtype != th.float32:
                raise RuntimeError("the native path is fp32 (use_fp16 is off in the reference defaults)")

    def plan(self, B, T, H, W, want_attn=False):
        key = (B, T, H, W, bool(want_attn))
        pl = self.plans.get(key)
        if pl is None:
            pl = Plan(self, B, T, H, W, bool(want_attn))
            pl.refresh_weights()
            self.plans[key] = pl
        elif pl._sig != pl.weight_signature():
            pl.refresh_weights()
        return pl

    def invalidate(self):
        """Parameters were rewritten behind torch's back: repack the conv weights on next use."""
        self.epoch += 1
        nat.param_epoch[0] += 1

    def forward(self, x, x0, timesteps, frame_indices, obs_mask, latent_mask, return_attn_weights=False):
        B, T, Cx, H, W = x.shape
        grad_needed = th.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.model.parameters()))
        # nn.Dropout is live in train() mode even without gradients (unet.py:166): the plan has no dropout stage
        if grad_needed or (self.model.training and self.model.dropout > 0):
            from ._autograd import unet_apply
            return unet_apply(self, x, x0, timesteps, frame_indices, obs_mask, latent_mask, return_attn_weights)
        pl = self.plan(B, T, H, W, return_attn_weights)
        pl.set_inputs(x.detach(), x0.detach(), timesteps, frame_indices, obs_mask, latent_mask)
        pl.launch()
        out = pl.out.clone()
        attns = None
        if return_attn_weights:
            # reference logs |mean over heads| per attention layer (rpe.py:128-131)
            attns = {"spatial": [a.mean(dim=1).abs() for a in pl.attn_s],
                     "temporal": [a.mean(dim=1).abs() for a in pl.attn_t], "mixed": []}
        return out, attns
