"""Developer aid: in-kernel timeline of the persistent level chains of the cfg-B sampler plan.

Needs devlib/liblfvdm_chainstamp.so (tools/build_chainstamp.sh: level_chain.hip with -DLFVDM_CHAIN_STAMP - thread 0 of
every workgroup stamps the 100 MHz s_memrealtime clock, common to all CUs, at the phase boundaries of every stage).
For every stage of every chain prints (microseconds, relative to the first stamp of the launch):
  beg    first workgroup entering the stage          end   last workgroup leaving it (flag published)
and medians over the workgroups that own work in the stage:
  pro    entry -> filter pieces issued + rows decoded (ready to poll)      poll  waiting for the producers' flags
  loop   activation pieces issued -> K loop done       red   LDS reduction        seam  slab store, ticket, ordered sum
  epi    bias / residual / store                       gn    fused GroupNorm      pub   drain + flag
usage: LFVDM_TUNE_CACHE=profiles/tune_cache_mi355x.json python tools/chain_stamps.py
"""
import ctypes as C
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_LIB_PATH", os.path.join(ROOT, "devlib", "liblfvdm_chainstamp.so"))
import torch as th  # noqa: E402

import bench  # noqa: E402
from improved_diffusion import _native as nat  # noqa: E402
from improved_diffusion._engine import Plan  # noqa: E402

NSTG, NWG, NS = 64, 256, 20


def med(v):
    return st.median(v) if v else float("nan")


def main():
    ch_n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = th.device("cuda")
    model, diffusion = bench.make_model_and_diffusion(ch_n, dev)
    B, T = 2, 20
    inputs = bench.synthetic_inputs(B, T, 0, dev)
    pl = Plan(model.native_engine(), B, T, 16, 16, False)
    pl.refresh_weights()
    pl.set_inputs(th.randn(B, T, 4, 16, 16, device=dev), inputs["x0"], th.tensor([500.0, 20.0], device=dev),
                  inputs["frame_indices"], inputs["obs_mask"], inputs["latent_mask"])
    pl.launch()
    pl.autotune()
    L = nat.lib()
    L.lfvdm_debug_chain_stamps.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * (NSTG * NWG * NS))()
    s = nat.stream()
    for _ in range(3):
        pl.launch()
    th.cuda.synchronize()
    for ci, ch in enumerate(pl.chains):
        fn, args = ch["step"]
        # a few back-to-back launches of the chain alone, the last one is read
        L.lfvdm_debug_chain_stamps(buf, 1)
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn(*args, s)
        th.cuda.synchronize()
        L.lfvdm_debug_chain_stamps(buf, 1)
        e0.record()
        fn(*args, s)
        e1.record()
        th.cuda.synchronize()
        L.lfvdm_debug_chain_stamps(buf, 0)
        ev_us = 1000.0 * e0.elapsed_time(e1)

        def at(stage, wg, i):
            return buf[(stage * NWG + wg) * NS + i]

        t0 = min(at(sg, w, 0) for sg in range(ch["n"]) for w in range(NWG) if at(sg, w, 0))
        print(f"chain {ci}: {ch['n']} stages, grid {ch['grid']}, event time {ev_us:.1f} us (stamped launch)")
        print(" stage kind items |   beg    end  span |  pro  poll  loop   red  seam   epi    gn   pub | first-A-landed | absolute: median poll end, median / last flag store")
        for sg in range(ch["n"]):
            wgs = [w for w in range(NWG) if at(sg, w, 0)]
            if not wgs:
                continue
            us = lambda x: (x - t0) / 100.0           # noqa: E731
            beg = min(us(at(sg, w, 0)) for w in wgs)
            end = max(us(max(at(sg, w, i) for i in range(NS))) for w in wgs)
            d = {k: [] for k in ("pro", "poll", "loop", "red", "seam", "epi", "gn", "pub", "fa")}
            for w in wgs:
                g = lambda i: at(sg, w, i)            # noqa: E731
                if g(16) and g(0):
                    d["pro"].append((g(16) - g(0)) / 100.0)
                if g(17) and g(16):
                    d["poll"].append((g(17) - g(16)) / 100.0)
                if g(2) and g(17):
                    d["loop"].append((g(2) - g(17)) / 100.0)
                if (g(15) or g(1)) and g(17):       # tile body: first activation piece landed; sample-local body: rows staged
                    d["fa"].append(((g(15) or g(1)) - g(17)) / 100.0)
                if g(3) and g(2):
                    d["red"].append((g(3) - g(2)) / 100.0)
                if g(6) and g(3):
                    d["seam"].append((g(6) - g(3)) / 100.0)
                if g(7) and (g(6) or g(3)):
                    d["epi"].append((g(7) - (g(6) or g(3))) / 100.0)
                if g(8) and g(7) and (g(12) or ch["kinds"][sg] == 2):
                    d["gn"].append((g(8) - g(7)) / 100.0)
                if g(18) and (g(8) or g(17)):
                    d["pub"].append((g(18) - (g(8) or g(17))) / 100.0)
            a17 = med([us(at(sg, w, 17)) for w in wgs if at(sg, w, 17)])
            a18 = [us(at(sg, w, 18)) for w in wgs if at(sg, w, 18)]
            a18m, a18x = med(a18), (max(a18) if a18 else float("nan"))
            kind = {0: "conv", 1: "gn", 2: "loc"}[ch["kinds"][sg]]
            stg = med([(at(sg, w, 4) - at(sg, w, 17)) / 100.0 for w in wgs if at(sg, w, 4) and at(sg, w, 17)])
            dma = med([(at(sg, w, 1) - at(sg, w, 4)) / 100.0 for w in wgs if at(sg, w, 4) and at(sg, w, 1)])
            kind = kind + (f" stage {stg:4.2f} dma-wait {dma:4.2f}" if ch["kinds"][sg] == 2 else "")
            print(f"  {sg:3d}  {kind:4s} {ch['items'][sg]:5d} | {beg:6.2f} {end:6.2f} {end - beg:5.2f} | "
                  f"{med(d['pro']):4.2f} {med(d['poll']):5.2f} {med(d['loop']):5.2f} {med(d['red']):5.2f} {med(d['seam']):5.2f} "
                  f"{med(d['epi']):5.2f} {med(d['gn']):5.2f} {med(d['pub']):5.2f} | {med(d['fa']):5.2f} | polled {a17:6.2f} published {a18m:6.2f} / {a18x:6.2f}")


if __name__ == "__main__":
    main()
