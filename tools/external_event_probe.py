#!/usr/bin/env python3
"""Does an EXTERNAL event recorded in the middle of a captured hipGraph order work on another stream that is issued
after ``graph.replay()``?  (Needed by the bucketed gradient exchange: a bucket's all-reduce on the side stream must start
when its last gradient has been written inside the replayed backward graph, not at the end of the graph.)

The graph is  [slow kernel A -> buf=1] -> record(ev) -> [slow kernel B -> buf2=1].  After the replay a side stream waits
on ev and copies buf, buf2.  Expected: the copy of buf sees 1 (ordered after A), and the side stream finishes BEFORE the
main stream (it did not wait for B) - the timestamps tell.
"""
import json
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-flexible-video-diffusion-modeling_amd"))
from improved_diffusion import _native as nat  # noqa: E402


def spin(x, iters):
    for _ in range(iters):
        x.mul_(1.0000001).add_(1e-9)


def main():
    dev = th.device("cuda", 0)
    big = th.ones(64 * 1024 * 1024, device=dev)
    buf = th.zeros(1024, device=dev)
    buf2 = th.zeros(1024, device=dev)
    out = th.zeros(1024, device=dev)
    out2 = th.zeros(1024, device=dev)
    res = {"torch": th.__version__}
    flags = nat.StreamFlags(1, dev)     # external events are refused on ROCm: a counter bumped by a kernel node instead
    side = th.cuda.Stream()
    s = th.cuda.Stream()
    s.wait_stream(th.cuda.current_stream())
    with th.cuda.stream(s):         # warm-up
        spin(big, 2)
    th.cuda.current_stream().wait_stream(s)
    th.cuda.synchronize()
    g = th.cuda.CUDAGraph()
    with th.cuda.graph(g):
        spin(big, 20)
        buf.fill_(1.0)
        flags.add(0)
        spin(big, 60)
        buf2.fill_(1.0)
    th.cuda.synchronize()
    ok_all = True
    rounds = []
    for it in range(5):
        buf.zero_(); buf2.zero_(); out.zero_(); out2.zero_()
        th.cuda.synchronize()
        t_side = th.cuda.Event(enable_timing=True)
        t_main = th.cuda.Event(enable_timing=True)
        t0 = th.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        with th.cuda.stream(side):
            flags.wait(0, it + 1, side)
            out.copy_(buf)
            out2.copy_(buf2)
            t_side.record(side)
        t_main.record()
        th.cuda.synchronize()
        a, b = float(out[0]), float(out2[0])
        ms_side, ms_main = t0.elapsed_time(t_side), t0.elapsed_time(t_main)
        rounds.append({"buf_seen": a, "buf2_seen": b, "side_done_ms": round(ms_side, 3), "main_done_ms": round(ms_main, 3)})
        ok_all &= (a == 1.0)
    res["rounds"] = rounds
    res["ordered_after_A"] = bool(ok_all)
    res["timed_out"] = flags.timed_out()
    res["overlaps_B"] = bool(all(r["side_done_ms"] < 0.7 * r["main_done_ms"] for r in rounds))
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
