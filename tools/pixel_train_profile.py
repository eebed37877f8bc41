#!/usr/bin/env python3
"""Pixel-space training steps for rocprofv3 (`rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/pixel_train_profile.py`)
or plain timing: the same leg bench.py reports as pixel.train.

    python3 tools/pixel_train_profile.py [--batch 1] [--rb 1] [--steps 6]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LFVDM_TUNE_CACHE", os.path.join(ROOT, "profiles", "tune_cache_mi355x.json"))

import torch as th  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rb", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=4)
    a = ap.parse_args()
    th.cuda.set_device(0)
    print(json.dumps(bench.bench_pixel_train(th.device("cuda", 0), a.steps, a.batch, a.rb, a.warmup)), flush=True)
