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
