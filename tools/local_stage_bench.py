"""Micro-benchmark of the SAMPLE-LOCAL chain stage (LFVDM_CHAIN_LOCAL, csrc/conv_local_body.h) against the split-K tile body
the persistent level chains used through round 5.

A chain of NST identical stages - 128 -> 128 3x3 convolution with the next layer's GroupNorm32 + FiLM + SiLU fused into the
epilogue, 40 samples (batch 2 x 20 frames) on 2x2 maps (M = 160) and 4x4 maps (M = 640) - is planned with lfvdm_chain_plan
and launched with lfvdm_level_chain, once per stage kind:
  tile   LFVDM_CHAIN_CONV, tile <1,1,4,1>, the split-K factor / LDS-DMA stages of the committed tune table (5 / 3 slices)
  local  LFVDM_CHAIN_LOCAL with 1 (and at M = 640 also 2) row tiles per item
Reports microseconds per stage (events around back-to-back launches, minimum over rounds), checks every kind against one
stand-alone lfvdm_conv_igemm launch per stage, and - LOCAL_BENCH_ONLY=<kind>:<H>[:rt] - runs a single variant so that a
rocprofv3 --pmc pass can attribute its counters (tools/refresh_profiles.sh, stage "localbench").
usage: python tools/local_stage_bench.py [out.json]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
import torch as th  # noqa: E402

from improved_diffusion import _native as nat  # noqa: E402

KZ_TABLE = [1, 2, 4, 8, 16, 3, 6, 5]
NST, N, CH, T = 8, 40, 128, 20


def tile_code(tile_id, kz, gl):
    return 1 + tile_id + 32 * KZ_TABLE.index(kz) + 256 * (gl - 1)


class Net:
    """NST stages on H x H maps: x_{i+1} = silu(film(GN32(conv3x3(x_i) + b)))."""

    def __init__(self, H, dev, seed=0):
        g = th.Generator(device="cpu").manual_seed(seed)
        self.H, self.M = H, N * H * H
        rn = lambda *s: th.randn(*s, generator=g)        # noqa: E731
        self.W = [(rn(CH, 9, CH) / (9 * CH) ** 0.5).to(dev) for _ in range(NST)]          # packed [Cout][tap][Cin]
        self.b = [(0.1 * rn(CH)).to(dev) for _ in range(NST)]
        self.gamma = [(1.0 + 0.1 * rn(CH)).to(dev) for _ in range(NST)]
        self.beta = [(0.1 * rn(CH)).to(dev) for _ in range(NST)]
        self.film = [(0.2 * rn(N // T, 2 * CH)).to(dev) for _ in range(NST)]
        self.x0 = rn(self.M, CH).to(dev)

    def bufs(self):
        return [self.x0] + [th.zeros(self.M, CH, device=self.x0.device) for _ in range(NST)]

    def conv_args(self, i, src, dst, a=None):
        a = a if a is not None else nat.ConvArgs()
        H = self.H
        a.src0, a.C0, a.C1, a.N, a.Hs, a.Ws, a.Ho, a.Wo = src.data_ptr(), CH, 0, N, H, H, H, H
        a.up, a.stride, a.ksize = 0, 1, 3
        a.W, a.bias, a.Cout = self.W[i].data_ptr(), self.b[i].data_ptr(), CH
        a.out, a.ldo, a.out_mode = dst.data_ptr(), CH, nat.OUT_ROWS          # (not written: gn_skip_raw)
        a.gn_gamma, a.gn_beta, a.gn_film = self.gamma[i].data_ptr(), self.beta[i].data_ptr(), self.film[i].data_ptr()
        a.gn_out, a.gn_film_ld, a.gn_film_div, a.gn_act, a.gn_skip_raw, a.gn_eps = dst.data_ptr(), 2 * CH, T, nat.ACT_SILU, 1, 1e-5
        return a

    def reference(self):
        """One stand-alone launch per stage (heuristic tile, no split-K workspace)."""
        L, s = nat.lib(), nat.stream()
        x = self.bufs()
        for i in range(NST):
            a = self.conv_args(i, x[i], x[i + 1])
            a.tune = 0
            nat.check(L.lfvdm_conv_igemm(C.byref(a), s), "lfvdm_conv_igemm")
        th.cuda.synchronize()
        return x[-1]

    def chain(self, kind, rt=1, code=0):
        L = nat.lib()
        x = self.bufs()
        stages = (nat.ChainStage * NST)()
        for i in range(NST):
            st = stages[i]
            st.kind = nat.CHAIN_LOCAL if kind == "local" else nat.CHAIN_CONV
            st.cfg = rt
            self.conv_args(i, x[i], x[i + 1], st.conv)
            st.conv.tune = code
        cap = 1 << 20
        deps = (C.c_int32 * cap)()
        used, ws_f, cnt_i = C.c_int64(), C.c_int64(), C.c_int64()
        n_flags, grid, lds = C.c_int32(), C.c_int32(), C.c_int32()
        nat.check(L.lfvdm_chain_plan(stages, NST, deps, cap, C.byref(used), C.byref(n_flags), C.byref(ws_f), C.byref(cnt_i),
                                     C.byref(grid), C.byref(lds)), "lfvdm_chain_plan")
        dev = self.x0.device
        ws = th.empty(max(1, ws_f.value), device=dev)
        cnt = th.zeros(max(1, cnt_i.value), dtype=th.int32, device=dev)
        for i in range(NST):
            cv = stages[i].conv
            cv.splitk_ws, cv.splitk_cnt = ws.data_ptr() + 4 * stages[i].ws_off, cnt.data_ptr() + 4 * stages[i].cnt_off
            cv.splitk_ws_floats, cv.splitk_cnt_ints = ws.numel() - stages[i].ws_off, cnt.numel() - stages[i].cnt_off
        stages_dev = th.frombuffer(bytearray(bytes(memoryview(stages))), dtype=th.uint8).to(dev)
        deps_dev = th.frombuffer(bytearray(bytes(memoryview(deps))[:4 * max(1, used.value)]), dtype=th.int32).to(dev)
        flags = th.zeros(max(1, n_flags.value), dtype=th.int32, device=dev)
        ctl = th.zeros(nat.CHAIN_CTL_INTS, dtype=th.int32, device=dev)
        keep = (x, ws, cnt, stages_dev, deps_dev, flags, ctl)
        args = (stages_dev.data_ptr(), NST, deps_dev.data_ptr(), flags.data_ptr(), ctl.data_ptr(), grid.value, lds.value, 2.0)
        info = dict(grid=grid.value, lds=lds.value, items=[s.n_items for s in stages], max_deps=max(s.dep_stride - 1 for s in stages))
        return args, keep, info


def time_chain(args, ctl, reps=50, rounds=5):
    L, s = nat.lib(), nat.stream()
    for _ in range(5):
        nat.check(L.lfvdm_level_chain(*args, s), "lfvdm_level_chain")
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(rounds):
        e0.record()
        for _ in range(reps):
            L.lfvdm_level_chain(*args, s)
        e1.record()
        e1.synchronize()
        best = min(best, 1000.0 * e0.elapsed_time(e1) / reps)
    assert int(ctl[nat.CHAIN_CTL_ABORT].item()) == 0, "a chain wait timed out"
    return best


def stamp_report(args, n_items):
    """Diagnostic build only (LFVDM_LIB_PATH=devlib/liblfvdm_chainstamp.so, tools/build_chainstamp.sh): medians over the
    workgroups of a stage of the in-kernel phase stamps (100 MHz clock), microseconds."""
    import statistics
    L, s = nat.lib(), nat.stream()
    if not hasattr(L, "lfvdm_debug_chain_stamps"):
        return None
    NSTG, NWG, NSL = 64, 256, 20
    L.lfvdm_debug_chain_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * (NSTG * NWG * NSL))()
    for _ in range(3):
        L.lfvdm_level_chain(*args, s)
    th.cuda.synchronize()
    L.lfvdm_debug_chain_stamps(buf, 1)
    L.lfvdm_level_chain(*args, s)
    th.cuda.synchronize()
    L.lfvdm_debug_chain_stamps(buf, 0)
    at = lambda sg, w, i: buf[(sg * NWG + w) * NSL + i]        # noqa: E731
    t0 = min(at(sg, w, 0) for sg in range(NST) for w in range(NWG) if at(sg, w, 0))
    med = lambda v: round(statistics.median(v), 2) if v else None      # noqa: E731
    rows = []
    for sg in range(NST):
        wgs = [w for w in range(NWG) if at(sg, w, 0)]
        d = {k: [] for k in ("pro", "poll", "act", "loop", "red", "epi", "gn", "pub")}
        for w in wgs:
            g = lambda i: at(sg, w, i)         # noqa: E731
            pairs = (("pro", 16, 0), ("poll", 17, 16), ("act", 1, 17), ("loop", 2, 1), ("red", 3, 2), ("epi", 7, 3), ("gn", 8, 7),
                     ("pub", 18, 8))
            for k, hi, lo in pairs:
                if g(hi) and g(lo):
                    d[k].append((g(hi) - g(lo)) / 100.0)
        pub = [(at(sg, w, 18) - t0) / 100.0 for w in wgs if at(sg, w, 18)]
        row = dict(stage=sg, workgroups=len(wgs), last_flag_us=round(max(pub), 2) if pub else None,
                   **{k: med(v) for k, v in d.items()})
        # flag hop of the stage's items: poll satisfied - the last of the producers' flag stores (items of the previous
        # stage with the same samples; one item per workgroup and stage, stages rotated by their item count)
        grid, n_it = 256, n_items[sg]
        if sg > 0 and n_it <= grid and n_items[sg - 1] == n_it:
            NS = CH // 16
            hops, spread = [], []
            for it in range(n_it):
                w = (it + sg * n_it) % grid
                prod = [((it // NS) * NS + sl + (sg - 1) * n_it) % grid for sl in range(NS)]
                tp = [at(sg - 1, pw, 18) for pw in prod if at(sg - 1, pw, 18)]
                if tp and at(sg, w, 17):
                    hops.append((at(sg, w, 17) - max(tp)) / 100.0)
                    spread.append((max(tp) - min(tp)) / 100.0)
            row["hop"] = med(hops)
            row["hop_max"] = round(max(hops), 2) if hops else None
            row["producer_spread"] = med(spread)
        rows.append(row)
    return rows


def main():
    dev = th.device("cuda")
    only = os.environ.get("LOCAL_BENCH_ONLY", "")
    out = {"stages_per_chain": NST, "layer": "128->128 3x3 + fused GroupNorm32/FiLM/SiLU, 40 samples", "variants": []}
    variants = []
    for H, kz in ((2, 5), (4, 3)):
        variants.append(("tile", H, 1, tile_code(6, kz, 2)))
        variants.append(("local", H, 1, 0))
        if H == 4:
            variants.append(("local", H, 2, 0))
    if only:
        parts = only.split(":")
        variants = [v for v in variants if v[0] == parts[0] and v[1] == int(parts[1]) and (len(parts) < 3 or v[2] == int(parts[2]))]
    for kind, H, rt, code in variants:
        net = Net(H, dev)
        ref = net.reference()
        args, keep, info = net.chain(kind, rt, code)
        nat.check(nat.lib().lfvdm_level_chain(*args, nat.stream()), "lfvdm_level_chain")
        th.cuda.synchronize()
        y = keep[0][-1]
        err = float((y - ref).abs().max())
        us = time_chain(args, keep[-1], reps=200 if only else 50)
        rec = dict(kind=kind, H=H, M=net.M, row_tiles=rt, us_per_chain=round(us, 2), us_per_stage=round(us / NST, 2),
                   max_abs_diff_vs_per_launch=err, out_absmax=float(ref.abs().max()), **info)
        if os.environ.get("LOCAL_BENCH_STAMPS") and kind == "local":
            rec["phase_stamps_us"] = stamp_report(args, info["items"])
        out["variants"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        assert err < 2e-4 * max(1.0, float(ref.abs().max())), f"{kind} H={H}: chain differs from the per-launch result by {err}"
    print(json.dumps(out))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
