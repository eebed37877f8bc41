#!/bin/bash
# Developer aid: A/B two builds of the library on the sampling bench (fresh tune cache each, alternating runs).
#   bash tools/ab_libs.sh <libA.so> <libB.so>
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for lib in "$1" "$2"; do
    export LFVDM_LIB_PATH=$PWD/$lib LFVDM_TUNE_CACHE_OUT=/tmp/tune_$(basename $lib).json LFVDM_TUNE_CACHE=
    python bench.py --steps 900 --warmup 50 --train-steps 6 --no-cpu --no-breakdown 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'train', d['train']['optimizer_steps_per_s'])"
  done
done
