"""Developer aid ("persistent level chain"): price of a fence-free device-wide dependency on MI355X.

Builds tools/grid_barrier_bench.hip with hipcc (gfx950) and runs it: a flat-counter barrier among 64 / 128 / 256
co-resident workgroups (relaxed agent-scope add + sc1 poll), the owners-only fan-in a split-K stage needs, and a data
hand-off (sc1 write-through stores -> sc1 loads / sc1 LDS-DMA) checked word by word under uneven load, with a plain-load
negative control.  Every spin in the kernels is bounded by a wall-clock timeout that raises an abort word.

usage: python tools/grid_barrier_bench.py [iters] > profiles/r05_grid_barrier.json
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "grid_barrier_bench.hip")
BIN = os.path.join(ROOT, "tools", "_bin", "grid_barrier_bench")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-o", BIN, SRC])
    return BIN


if __name__ == "__main__":
    exe = build()
    iters = sys.argv[1] if len(sys.argv) > 1 else "2000"
    sys.exit(subprocess.call([exe, iters]))
