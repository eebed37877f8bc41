"""Developer aid: in-kernel phase table of the implicit-GEMM launches of the cfg-B sampler plan.

Needs the diagnostic build devlib/liblfvdm_stamp.so (conv_igemm.hip compiled with -DLFVDM_STAMP: thread 0 of every
workgroup stamps the 100 MHz s_memrealtime clock, common to all CUs, at the phase boundaries).  For every distinct
launch shape whose template instance matches FILTER (default "1141" = <1,1,4,1>) prints, in microseconds:
  skew   first -> last workgroup entering the kernel
  pro    kernel entry -> first DMA pieces issued (row decode, descriptors)
  loop   K loop (DMA wait + MFMA)
  red    cross-k-group LDS reduction
  slab   slab stores + vmcnt(0) + barrier           (split-K only)
  rel    release fence                               (split-K only)
  tick   ticket atomic round trip                    (split-K only)
  acq    acquire fence of the last arriver           (split-K only)
  sum    ordered slab sum                            (split-K only)
  epi    bias / residual / raw store
  gn     fused GroupNorm epilogue
  span   first entry -> last stamp of any workgroup; `event` = HIP-event time per launch of the same launch
usage: LFVDM_TUNE_CACHE=profiles/tune_cache_mi355x.json python tools/conv_phase_stamps.py [FILTER] [ch]
"""
import ctypes as C
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_LIB_PATH", os.path.join(ROOT, "devlib", "liblfvdm_stamp.so"))
import torch as th  # noqa: E402

import bench  # noqa: E402
from improved_diffusion import _native as nat  # noqa: E402
from improved_diffusion._engine import Plan  # noqa: E402

NS, NW = 16, 2048


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else "1141"
    ch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = th.device("cuda")
    model, diffusion = bench.make_model_and_diffusion(ch, dev)
    B, T = 2, 20
    inputs = bench.synthetic_inputs(B, T, 0, dev)
    pl = Plan(model.native_engine(), B, T, 16, 16, False)
    pl.refresh_weights()
    pl.set_inputs(th.randn(B, T, 4, 16, 16, device=dev), inputs["x0"], th.tensor([500.0, 20.0], device=dev),
                  inputs["frame_indices"], inputs["obs_mask"], inputs["latent_mask"])
    pl.autotune()
    L = nat.lib()
    L.lfvdm_debug_stamps.argtypes = [C.c_void_p]
    s = nat.stream()
    pl.launch()
    th.cuda.synchronize()
    seen = set()
    buf = (C.c_ulonglong * (NS * NW))()
    print("inst    M Cin Cout k s2 gn kz wgs |  skew   pro  loop   red  slab   rel  tick   acq   sum   epi    gn  1stc gn:wr gn:st gn:ba gn:ap | span event")
    for i, (fn, args) in enumerate(pl.steps):
        if fn is not L.lfvdm_conv_igemm:
            continue
        a = args[0]._obj
        nt, nw = C.c_int(), C.c_int()
        L.lfvdm_conv_igemm_config(C.byref(a), C.byref(nt), C.byref(nw))
        inst = str(nt.value)
        if flt != "all" and inst != flt:
            continue
        key = nat.tune_key(a) + (a.tune,)
        if key in seen:
            continue
        seen.add(key)
        t = a.tune - 1
        kz = (1, 2, 4, 8, 16, 3, 6, 5)[(t >> 5) & 7] if a.tune > 0 else 1      # 16 = tail split
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        for _ in range(5):
            fn(*args, s)
        e0.record()
        for _ in range(20):
            fn(*args, s)
        e1.record()
        th.cuda.synchronize()
        ev_us = e0.elapsed_time(e1) * 1000 / 20
        L.lfvdm_debug_stamps_clear()
        # a realistic predecessor (another small kernel) in front, then the launch under test
        pl.steps[i - 1][0](*pl.steps[i - 1][1], s)
        fn(*args, s)
        th.cuda.synchronize()
        L.lfvdm_debug_stamps(buf)
        rows = [[buf[w * NS + j] for j in range(NS)] for w in range(NW)]
        rows = [r for r in rows if r[0]]
        # stamps older than the workgroup's own entry were left by the predecessor launch: not passed in this one
        rows = [[x if x >= r[0] else 0 for x in r] for r in rows]
        t0 = min(r[0] for r in rows)

        def med(i, j, sel=lambda r: True):
            """median over workgroups of stamp j - stamp i; None where a workgroup did not pass both"""
            v = [r[j] - r[i] for r in rows if sel(r) and r[i] and r[j] and r[j] >= r[i]]
            return st.median(v) / 100.0 if v else None

        split = any(r[4] for r in rows)
        last = (lambda r: r[6] > 0) if split else (lambda r: True)
        skew = (max(r[0] for r in rows) - t0) / 100.0
        span = (max(max(r) for r in rows) - t0) / 100.0
        cols = [skew, med(0, 1), med(1, 2), med(2, 3)]
        if split:
            cols += [med(3, 4), med(4, 9), med(9, 10), med(10, 11, last), med(5, 6, last), med(6, 7, last), med(7, 8, last)]
        else:
            cols += [None] * 5 + [med(3, 7), med(7, 8)]
        # finer split: first chunk landed (cold filters), GroupNorm: tile rewrite / statistics / last barrier / apply
        cols += [med(1, 15), med(7, 12, last), med(12, 13, last), med(13, 14, last), med(14, 8, last)]
        M = a.N * a.Ho * a.Wo
        print(f"{inst} {M:5d} {a.C0 + a.C1:3d} {a.Cout:4d} {a.ksize} {a.s2C0 + a.s2C1:3d} {int(bool(a.gn_out)):2d} {kz:2d} {len(rows):4d} | "
              + " ".join("    -" if c is None else f"{c:5.2f}" for c in cols) + f" | {span:5.2f} {ev_us:5.2f}")
        if os.environ.get("STAMP_DUMP"):
            for r in sorted(rows, key=lambda r: r[0])[:int(os.environ["STAMP_DUMP"])]:
                print("   ", [(x - t0) / 100.0 if x else None for x in r[:12]])


if __name__ == "__main__":
    main()
