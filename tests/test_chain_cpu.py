"""Host-side planning of the persistent level chain (lfvdm_chain_plan, csrc/level_chain.hip): work items, tile flags and
dependency lists are computed without a GPU - checked here against the tile arithmetic of the stand-alone kernel, restated
in Python.  (The kernel itself is covered by tests/test_chain_gpu.py: bitwise equal to the per-launch plan.)"""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
from improved_diffusion import _native as nat  # noqa: E402

if not os.path.exists(nat.LIB_PATH):
    pytest.skip("native library not built", allow_module_level=True)

KZ_TABLE = [1, 2, 4, 8, 16, 3, 6, 5]


def code(tile_id, kz, gl):
    """tune code of (tile id, 32-channel chunks, split-K factor, LDS-DMA stages) - include/lfvdm_hip.h, lfvdm_conv_args.tune"""
    return 1 + tile_id + 32 * KZ_TABLE.index(kz) + 256 * (gl - 1)


def conv_stage(src, out, Cin, Cout, N, H, tune, gn_out=0, skip_raw=0, res=0, src1=0, C1=0, ksize=3):
    st = nat.ChainStage()
    st.kind = nat.CHAIN_CONV
    a = st.conv
    a.src0, a.C0, a.src1, a.C1 = src, Cin, src1 or None, C1
    a.N, a.Hs, a.Ws, a.Ho, a.Wo, a.stride, a.ksize, a.up = N, H, H, H, H, 1, ksize, 0
    a.W, a.bias, a.Cout, a.out, a.ldo, a.out_mode, a.tune = 0x10, 0x20, Cout, out, Cout, nat.OUT_ROWS, tune
    if res:
        a.res, a.ldr = res, Cout
    if gn_out:
        a.gn_gamma, a.gn_beta, a.gn_out, a.gn_film_div, a.gn_skip_raw, a.gn_eps = 0x30, 0x40, gn_out, 1, skip_raw, 1e-5
    return st


def gn_stage(src0, src1, C0, C1, N, P, out):
    st = nat.ChainStage()
    st.kind = nat.CHAIN_GN
    g = st.gn
    g.src0, g.src1, g.C0, g.C1, g.N, g.P, g.gamma, g.beta, g.eps, g.act, g.out = src0, src1 or None, C0, C1, N, P, 0x30, 0x40, 1e-5, 1, out
    return st


def plan(stages):
    L = nat.lib()
    arr = (nat.ChainStage * len(stages))(*stages)
    cap = 1 << 18
    deps = (C.c_int32 * cap)()
    used, ws, cnt = C.c_int64(), C.c_int64(), C.c_int64()
    nfl, grid, lds = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.lfvdm_chain_plan(arr, len(stages), deps, cap, C.byref(used), C.byref(nfl), C.byref(ws), C.byref(cnt), C.byref(grid),
                            C.byref(lds))
    return rc, arr, list(deps[:used.value]), nfl.value, ws.value, cnt.value, grid.value, lds.value


def item_deps(st, deps, item):
    row = deps[st.dep_base + item * st.dep_stride: st.dep_base + (item + 1) * st.dep_stride]
    return sorted(row[1:1 + row[0]])


def test_resblock_chain_dependencies_follow_the_tile_arithmetic():
    """Two 128 -> 128 3x3 convolutions on 2x2 maps of 40 samples (M = 160: 5 row tiles x 4 filter tiles of the <1,1,4,1>
    instance) and a GroupNorm over the concat of both outputs.  A conv work item (row tile, filter tile, K slice) must wait
    for exactly the producer tiles that hold its rows and the channel chunks of its K slice (+ its residual tile); a
    GroupNorm item for the tiles that hold its sample and its 64 channels."""
    N, H, C0 = 40, 2, 128
    A, B_, Cb, D, E = 0x100000, 0x200000, 0x300000, 0x400000, 0x500000
    s0 = conv_stage(A, B_, C0, C0, N, H, code(6, 5, 2), gn_out=Cb, skip_raw=0)        # raw -> B_, normalised -> Cb
    s1 = conv_stage(Cb, D, C0, C0, N, H, code(6, 6, 3), res=B_)
    s2 = gn_stage(D, B_, C0, C0, N, H * H, E)
    rc, st, deps, nflags, ws, cnt, grid, lds = plan([s0, s1, s2])
    assert rc == 0
    MT, NT2, BM, BN, taps, chunks = 5, 4, 32, 32, 9, 4
    assert [s.n_flags for s in st] == [20, 20, N * 4] and nflags == 40 + 160
    assert st[0].kz == 5 and st[0].nt2 == 4 and st[0].n_items == 8 * -(-100 // 8) and st[1].n_items == 8 * -(-120 // 8)
    assert st[0].cfg == 0 and st[1].cfg == 2            # <1,1,4,1>, one-source form, 2 / 3 stages
    assert ws == (20 * 5 + 20 * 6) * BM * BN and cnt == 40 and st[1].ws_off == 20 * 5 * BM * BN and st[1].cnt_off == 20
    assert grid == 160 and lds == 4 * 3 * 64 * 32 * 4          # the GroupNorm stage has the most work items (40 samples x 4)
    # stage 0 reads only buffers written before the chain
    assert all(not item_deps(st[0], deps, i) for i in range(st[0].n_items))
    # stage 1: restate the kernel's XCD-aware item map and K-slice split
    KZ, total = 6, 120
    per = st[1].n_items // 8
    NK = taps * chunks
    seen = 0
    for item in range(st[1].n_items):
        Lidx = (item & 7) * per + (item >> 3)
        if Lidx >= total:
            assert item_deps(st[1], deps, item) == []
            continue
        pair, bx = divmod(Lidx, MT)
        by, kz = divmod(pair, KZ)
        zb, ze = NK * kz // KZ, NK * (kz + 1) // KZ
        cis = sorted({k // taps for k in range(zb, ze)})
        want = sorted({st[0].flag_base + ci * MT + bx for ci in cis} | {st[0].flag_base + by * MT + bx})   # operand tiles + residual tile
        assert item_deps(st[1], deps, item) == want, (item, bx, by, kz)
        seen += 1
    assert seen == total
    # GroupNorm items: (sample n, 64-channel block j) of the concat (D | B_): rows of sample n -> row tile n // 8
    for item in range(st[2].n_items):
        n, j = divmod(item, 4)
        prod = st[1] if j < 2 else st[0]
        cols = [2 * (j % 2), 2 * (j % 2) + 1]
        assert item_deps(st[2], deps, item) == sorted(prod.flag_base + c * MT + n // 8 for c in cols)


def test_a_buffer_written_twice_or_read_before_written_is_refused():
    """There is no launch boundary inside a chain: a stage that overwrites a buffer an earlier stage read or wrote cannot be
    ordered - the planner refuses, the caller keeps one launch per stage."""
    N, H, Cc = 40, 2, 128
    A, B_, Cb = 0x100000, 0x200000, 0x300000
    t = code(6, 5, 2)
    rc = plan([conv_stage(A, B_, Cc, Cc, N, H, t), conv_stage(B_, A, Cc, Cc, N, H, t)])[0]         # overwrites its producer's input
    assert rc == 3
    rc = plan([conv_stage(A, B_, Cc, Cc, N, H, t), conv_stage(B_, Cb, Cc, Cc, N, H, t), conv_stage(Cb, B_, Cc, Cc, N, H, t)])[0]
    assert rc == 3
    assert plan([conv_stage(A, B_, Cc, Cc, N, H, t), conv_stage(B_, Cb, Cc, Cc, N, H, t)])[0] == 0


def test_only_the_tile_family_of_the_chain_kernel_is_accepted():
    L = nat.lib()
    N, H, Cc = 40, 4, 128
    ok = conv_stage(0x1000, 0x2000, Cc, Cc, N, H, code(6, 3, 2))
    assert L.lfvdm_chain_conv_ok(C.byref(ok.conv)) == 0
    for bad in (code(3, 3, 2), code(6, 16, 2), 0, code(6, 3, 2) + 16):        # 8-wave tile, tail split, heuristic, 64-channel chunks
        st = conv_stage(0x1000, 0x2000, Cc, Cc, N, H, bad)
        assert L.lfvdm_chain_conv_ok(C.byref(st.conv)) != 0, bad
    nchw = conv_stage(0x1000, 0x2000, Cc, 4, N, H, code(6, 3, 2))
    nchw.conv.out_mode = nat.OUT_NCHW
    assert L.lfvdm_chain_conv_ok(C.byref(nchw.conv)) != 0
    assert L.lfvdm_chain_gn_ok(128, 128, 40, 16) == 0 and L.lfvdm_chain_gn_ok(128, 0, 40, 1024) != 0
    assert L.lfvdm_chain_gn_ok(96, 0, 40, 16) != 0           # 96 channels: not a multiple of 64


def test_concat_operand_written_half_by_half():
    """The decoder's first GroupNorm over concat(h, skip) evaluated half by half: the producer GEMM writes the left 128 columns
    of the consumer's 256-column operand from its epilogue (gn_gw / gn_ld), a free-standing GroupNorm stage the right 128
    (out_base / out_col).  The consumer's K slices must wait for the GEMM's tiles where they read the left columns and for
    the GroupNorm's items where they read the right ones; the free-standing stage goes to the top workgroups of the grid."""
    N, H, Ch = 40, 2, 128
    X, RAW, ACT, SKIP, OUT = 0x100000, 0x200000, 0x300000, 0x400000, 0x500000
    prod = conv_stage(X, RAW, Ch, Ch, N, H, code(6, 5, 2), gn_out=ACT)
    prod.conv.gn_gw, prod.conv.gn_ld = 8, 2 * Ch
    part = gn_stage(SKIP, 0, Ch, 0, N, H * H, ACT + 4 * Ch)
    part.gn.cg, part.gn.ldo, part.gn.out_base, part.gn.out_col = 8, 2 * Ch, ACT, Ch
    cons = conv_stage(ACT, OUT, 2 * Ch, Ch, N, H, code(6, 6, 2))
    rc, st, deps, nflags, ws, cnt, grid, lds = plan([part, prod, cons])
    assert rc == 0
    MT, taps, KZ = 5, 9, 6
    assert st[0].n_items == N * 2 and st[0].wg_off == grid - st[0].n_items and st[1].wg_off == 0 and st[2].wg_off == 0
    assert all(not item_deps(st[0], deps, i) for i in range(st[0].n_items))
    per, total, NK = st[2].n_items // 8, MT * 4 * KZ, taps * 8
    for item in range(st[2].n_items):
        Lidx = (item & 7) * per + (item >> 3)
        if Lidx >= total:
            continue
        pair, bx = divmod(Lidx, MT)
        by, kz = divmod(pair, KZ)
        want = set()
        for ci in sorted({k // taps for k in range(NK * kz // KZ, NK * (kz + 1) // KZ)}):
            if ci < 4:          # left half: the producer GEMM's tile (row tile bx, filter tile ci)
                want.add(st[1].flag_base + ci * MT + bx)
            else:               # right half: the GroupNorm items (sample n, 64-column block j) of the tile's 8 samples
                j = (32 * (ci - 4)) // 64
                want |= {st[0].flag_base + n * 2 + j for n in range(8 * bx, 8 * bx + 8)}
        assert item_deps(st[2], deps, item) == sorted(want), (item, bx, by, kz)
    # the same columns written twice are refused; disjoint halves of one buffer are not
    again = gn_stage(SKIP, 0, Ch, 0, N, H * H, ACT + 4 * Ch)
    again.gn.cg, again.gn.ldo, again.gn.out_base, again.gn.out_col = 8, 2 * Ch, ACT, Ch
    assert plan([part, again])[0] == 3


def local_item_deps(st, deps, item):
    """The lists of a LOCAL item: [n0, n1, n2, n3, r0, r1 | operand producers per wave | residual producers per row tile] ->
    the four per-wave lists, the residual producers merged into the list of the wave that finishes their row tile."""
    row = deps[st.dep_base + item * st.dep_stride: st.dep_base + (item + 1) * st.dep_stride]
    lists, off = [], 6
    for w in range(6):
        lists.append(list(row[off:off + row[w]]))
        off += row[w]
    return [sorted(lists[w] + (lists[4 + w] if w < 2 else [])) for w in range(4)]


def local_stage(src, out, Cin, Cout, N, Hs, Ho, rt, gn_out=0, skip_raw=0, res=0, stride=1, up=0, s2=(0, 0, 0, 0), ksize=3):
    st = conv_stage(src, out, Cin, Cout, N, Hs, 0, gn_out=gn_out, skip_raw=skip_raw, res=res, ksize=ksize)
    st.kind, st.cfg = nat.CHAIN_LOCAL, rt
    a = st.conv
    a.Ho, a.Wo, a.stride, a.up = Ho, Ho, stride, up
    if s2[0]:
        a.s2src0, a.s2C0, a.s2src1, a.s2C1, a.W2, a.bias2 = s2[0], s2[1], s2[2] or None, s2[3], 0x50, 0x60
    return st


def test_sample_local_stages_items_flags_dependencies_and_rotation():
    """LFVDM_CHAIN_LOCAL (csrc/conv_local_body.h): item = (row group of 16 * rt rows = whole samples, slice of 16 filters),
    id = row group * slices + slice, flag = base + slice * row groups + row group.  An item reads ALL channels of its samples'
    source pixels (a stride-2 stage: the 4x larger source maps), the rows of its own tile of the 1x1 skip segment and its
    residual tile; consecutive LOCAL stages are rotated over the whole grid; the LDS request is front region + filter slice."""
    N, Ch = 40, 128
    X, A1, R1, A2, OUT, A3 = 0x100000, 0x200000, 0x300000, 0x400000, 0x500000, 0x600000
    s0 = local_stage(X, R1, Ch, Ch, N, 4, 2, 1, gn_out=A1, stride=2)                        # 4x4 -> 2x2: raw R1 + normalised A1
    s1 = local_stage(A1, 0x700000, Ch, Ch, N, 2, 2, 1, gn_out=A2, skip_raw=1)              # conv1 of a ResBlock
    s2 = local_stage(A2, OUT, Ch, Ch, N, 2, 2, 1, res=R1, gn_out=A3)                         # conv2 + residual
    s3 = local_stage(A3, 0x800000, Ch, Ch, N, 2, 4, 2, up=1, s2=(0, 0, 0, 0))                # nearest-2x + conv: 4x4, two row tiles
    rc, st, deps, nflags, ws, cnt, grid, lds = plan([s0, s1, s2, s3])
    assert rc == 0 and ws == 0 and cnt == 0
    NS = Ch // 16
    assert [s.n_items for s in st] == [10 * NS, 10 * NS, 10 * NS, 20 * NS] and [s.nt2 for s in st] == [NS] * 4
    assert grid == 256 and [s.wg_off for s in st] == [0, 80, 160, 240]
    front = lambda rows_in, rows, C2=0: 2048 + (rows_in + 1) * Ch + rows * C2          # noqa: E731
    assert [s.kz for s in st] == [front(64, 16), front(16, 16), front(16, 16), front(8, 32)]
    assert lds == 4 * (front(64, 16) + 16 * 5 * 256)          # filter rows in whole 1 KiB pieces: 9 * 128 floats -> 5 pieces
    assert all(local_item_deps(st[0], deps, i) == [[], [], [], []] for i in range(st[0].n_items))
    # one list per WAVE: wave w stages and multiplies channels [32 w, 32 w + 32) = filter slices 2 w, 2 w + 1 of the producer
    # (same 4 samples: row group rg); wave 0, which finishes the item's row tile, also waits for its residual tile
    for k, prod, resprod in ((1, st[0], None), (2, st[1], st[0])):
        MT = 10
        for item in range(st[k].n_items):
            rg, sl = divmod(item, NS)
            want = [sorted({prod.flag_base + c * MT + rg for c in (2 * w, 2 * w + 1)} |
                           ({resprod.flag_base + sl * MT + rg} if resprod is not None and w == 0 else set())) for w in range(4)]
            assert local_item_deps(st[k], deps, item) == want, (k, item)
    # the upsampling stage: 32 output rows = 2 samples = 8 source rows: half a producer row group (16 rows = 4 samples)
    for item in range(st[3].n_items):
        rg, sl = divmod(item, NS)
        assert local_item_deps(st[3], deps, item) == [sorted(st[2].flag_base + c * 10 + rg // 2 for c in (2 * w, 2 * w + 1))
                                                      for w in range(4)]
    # a grid cap (what lfvdm_chain_capacity reports on a smaller device) bounds the plan; rotation follows it
    L = nat.lib()
    arr = (nat.ChainStage * 4)(s0, s1, s2, s3)
    d2 = (C.c_int32 * (1 << 16))()
    used, wsv, cntv, nfl, g, ldsv = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32(96), C.c_int32()
    assert L.lfvdm_chain_plan(arr, 4, d2, 1 << 16, C.byref(used), C.byref(nfl), C.byref(wsv), C.byref(cntv), C.byref(g), C.byref(ldsv)) == 0
    assert g.value == 96 and [s.wg_off for s in arr] == [0, 80, 64, 48]


def test_sample_local_stage_acceptance():
    L = nat.lib()
    ok = local_stage(0x1000, 0x2000, 128, 128, 40, 2, 2, 1)
    assert L.lfvdm_chain_local_ok(C.byref(ok.conv), 1) == 0 and L.lfvdm_chain_local_ok(C.byref(ok.conv), 2) == 0
    assert L.lfvdm_chain_local_ok(C.byref(ok.conv), 3) != 0
    wide = local_stage(0x1000, 0x2000, 256, 128, 40, 2, 2, 1)             # 16 x 2304 filters + 17 rows of 256: over the LDS
    assert L.lfvdm_chain_local_ok(C.byref(wide.conv), 1) != 0
    big = local_stage(0x1000, 0x2000, 128, 128, 40, 8, 8, 1)              # 64-pixel maps: a sample does not fit a row tile
    assert L.lfvdm_chain_local_ok(C.byref(big.conv), 1) != 0
    odd = local_stage(0x1000, 0x2000, 96, 128, 40, 2, 2, 1)               # 96 channels: rows of the swizzled image are 64-float multiples
    assert L.lfvdm_chain_local_ok(C.byref(odd.conv), 1) != 0
    bad = local_stage(0x1000, 0x2000, 128, 128, 40, 4, 4, 1, stride=2)    # inconsistent geometry
    assert L.lfvdm_chain_local_ok(C.byref(bad.conv), 1) != 0
    mixed = plan([local_stage(0x1000, 0x2000, 128, 128, 40, 2, 2, 1, gn_out=0x3000),
                  conv_stage(0x3000, 0x4000, 128, 128, 40, 2, code(6, 5, 2)),
                  local_stage(0x4000, 0x5000, 128, 128, 40, 2, 2, 1)])
    assert mixed[0] == 0 and [s.kind for s in mixed[1]] == [nat.CHAIN_LOCAL, nat.CHAIN_CONV, nat.CHAIN_LOCAL]
    # the tile stage waits for the LOCAL producer's (row group, slice) flags that cover its rows and channel chunks ...
    st, deps = mixed[1], mixed[2]
    some = [item_deps(st[1], deps, i) for i in range(st[1].n_items)]
    assert any(d for d in some) and all(st[0].flag_base <= f < st[0].flag_base + st[0].n_flags for d in some for f in d)
    # ... and the LOCAL consumer for the tile stage's 32 x 32 tiles: wave w for column tile w, the row tile of its 4 samples
    for item in range(st[2].n_items):
        rg = item // 8
        assert local_item_deps(st[2], deps, item) == [[st[1].flag_base + w * 5 + rg // 2] for w in range(4)]


def test_cfgB_plan_becomes_two_sample_local_chains_with_a_skip_side():
    """Host side of round 6 on the headline configuration (BASELINE.json configs[1]; the launch plan is built on CPU tensors,
    nothing is launched): the 4x4 / 2x2 levels become two chains; every convolution stage is sample-local; the decoder's
    four convolutions over cat([h, skip]) (unet.py:460, :152-155) are split, their skip side - GroupNorm of the skip tensor,
    then the skip half of the convolution - are the chain's first stages, marked `side`, on the reserved top 80 workgroups;
    taken apart again the plan has its 102 launches back (94 + the 8 of the four splits) in a legal order."""
    import torch as th
    sys.path.insert(0, ROOT)
    import bench
    from improved_diffusion import _engine as eng
    model, _ = bench.make_model_and_diffusion(64, th.device("cpu"))
    pl = eng.Plan(eng.Engine(model), 2, 20, 16, 16, False, time_steps=1000)
    L = nat.lib()
    n_launches = len(pl.steps)
    for fn, args in pl.steps:                      # (stand-in for the autotuner: a chain-legal tile code for the tile stages)
        if fn is L.lfvdm_conv_igemm and args[0]._obj.N * args[0]._obj.Ho * args[0]._obj.Wo <= 640:
            args[0]._obj.tune = code(6, 4, 2)
    assert pl.build_chains() == 2 and len(pl.steps) == n_launches - 28 + 2
    enc, dec = pl.chains
    assert enc["kinds"] == [nat.CHAIN_LOCAL] * 8 and enc["grid"] == 256
    assert dec["n"] == 20 and dec["kinds"][:8] == [nat.CHAIN_GN, nat.CHAIN_LOCAL] * 4 and set(dec["kinds"][8:]) == {nat.CHAIN_LOCAL}
    side = [st.side for _, st in dec["run"]]
    assert side == [1] * 8 + [0] * 12 and not any(st.side for _, st in enc["run"])
    # the skip halves are plain convolutions into a partial tensor that the decoder halves take as their residual
    outs = {st.conv.out for _, st in dec["run"][:8] if st.kind == nat.CHAIN_LOCAL}
    users = [st for _, st in dec["run"][8:] if st.conv.res in outs]
    assert len(outs) == 4 and len(users) == 4 and all(st.conv.gn_out and st.conv.gn_film for st in users)
    assert all(not (st.conv.bias or st.conv.gn_out or st.conv.res) for _, st in dec["run"][:8] if st.kind == nat.CHAIN_LOCAL)
    assert sum(1 for e in pl.packs if len(e) == 4) == 8              # two packed halves per split convolution
    pl.disable_chains()
    assert len(pl.steps) == n_launches and not pl.chains
    order = [id(s) for s in pl.steps]
    for step, waits_for in pl.skip_side:                             # producer before consumer in the per-launch order
        assert id(step) in order
