#!/bin/bash
# One gpurun call of the development loop: graph-event probe, GPU tests, default bench, 2-rank rehearsal.
# A step that was killed at its limit stops the chain (no further GPU step in the same call).
OUT=gpurun_out/r2
mkdir -p $OUT
step() {   # step <seconds> <log> <cmd...>
    local lim=$1 log=$2; shift 2
    timeout -k 10 $lim "$@" > $OUT/$log 2>&1
    local rc=$?
    echo "[$log] rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi
    return 0
}
step 120 extev.json python tools/external_event_probe.py
cat $OUT/extev.json | tail -3
step 900 gpu_tests.log python -m pytest tests -m gpu -q -x --timeout 600
tail -15 $OUT/gpu_tests.log
[ "$1" = "tests" ] && exit 0
step 400 bench1.json python bench.py
tail -c 1500 $OUT/bench1.json
step 300 bench2.json python bench.py --gpus 2 --steps 200 --train-steps 10
tail -c 1200 $OUT/bench2.json
