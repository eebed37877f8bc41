#!/bin/bash
# round-3 development step: conv parity tests, phase stamps (sc1 seam vs fenced seam), sampling A/B
OUT=gpurun_out/r3
mkdir -p $OUT
export LFVDM_TUNE_CACHE=$PWD/profiles/tune_cache_mi355x.json
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -m gpu -q -x --timeout 300 -k "conv or forward or resblock or attention" > $OUT/t1.log 2>&1; rc=$?
tail -5 $OUT/t1.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python tools/conv_phase_stamps.py all > $OUT/stamps_sc1.txt 2>&1 || exit 1
LFVDM_LIB_PATH=$PWD/devlib/liblfvdm_stampf.so timeout -k 10 200 python tools/conv_phase_stamps.py all > $OUT/stamps_fence.txt 2>&1 || exit 1
B="python bench.py --steps 900 --warmup 50 --train-steps 0 --pixel-steps 0 --long-video-windows 0 --no-cpu --no-breakdown"
for rep in 1 2; do
  timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])"
  HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('devkernarg=1', d['value'])"
  HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('devkernarg=0', d['value'])"
done
