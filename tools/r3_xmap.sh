#!/bin/bash
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
rm -f $OUT/tune_a.json $OUT/tune_b.json
LFVDM_TUNE_CACHE=$OUT/tune_a.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_a.json timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_sampler_gpu.py -m gpu -q -x --timeout 300 -k "conv or forward or sampler or graph or fused" > $OUT/t_xmap.log 2>&1; rc=$?
tail -3 $OUT/t_xmap.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --steps 900 --warmup 50 --train-steps 0 --pixel-steps 0 --long-video-windows 0 --no-cpu"
for rep in 1 2; do
  LFVDM_TUNE_CACHE=$OUT/tune_a.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_a.json timeout -k 10 300 $B 2>/dev/null > $OUT/b_xmap.json; python -c "import json; d=json.loads(open('$OUT/b_xmap.json').read().strip().splitlines()[-1]); print('xcd map', d['value'], d['breakdown']['launches'], d['breakdown']['kernels'].get('conv_igemm_kernel<1,1,4,1>'), d['breakdown']['kernels'].get('lfvdm_conv_in'))"
  LFVDM_CONV_NO_XCD_MAP=1 LFVDM_TUNE_CACHE=$OUT/tune_b.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_b.json timeout -k 10 300 $B 2>/dev/null > $OUT/b_noxmap.json; python -c "import json; d=json.loads(open('$OUT/b_noxmap.json').read().strip().splitlines()[-1]); print('plain map', d['value'], d['breakdown']['launches'], d['breakdown']['kernels'].get('conv_igemm_kernel<1,1,4,1>'))"
done
