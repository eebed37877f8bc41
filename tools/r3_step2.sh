#!/bin/bash
# round-3 development step: conv parity tests, phase stamps, sampling bench
OUT=gpurun_out/r3
mkdir -p $OUT
export LFVDM_TUNE_CACHE=$PWD/profiles/tune_cache_mi355x.json
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_sampler_gpu.py -m gpu -q -x --timeout 300 > $OUT/t2.log 2>&1; rc=$?
tail -3 $OUT/t2.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python tools/conv_phase_stamps.py all > $OUT/stamps_2.txt 2>&1 || exit 1
B="python bench.py --steps 900 --warmup 50 --train-steps 0 --pixel-steps 0 --long-video-windows 0 --no-cpu --no-breakdown"
for rep in 1 2; do
  timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cached tune', d['value'])"
done
LFVDM_TUNE_CACHE= LFVDM_TUNE_CACHE_OUT=$PWD/$OUT/tune_new.json timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fresh tune', d['value'])"
