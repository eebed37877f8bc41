#!/bin/bash
# round-3 development step: fused spatial attention tests + sampling A/B
OUT=gpurun_out/r3
mkdir -p $OUT
export LFVDM_TUNE_CACHE=$PWD/$OUT/tune_new.json LFVDM_TUNE_CACHE_OUT=$PWD/$OUT/tune_new.json
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_sampler_gpu.py -m gpu -q -x --timeout 300 -k "spatial or forward or sampler or graph" > $OUT/t3.log 2>&1; rc=$?
tail -5 $OUT/t3.log
[ $rc -ne 0 ] && exit 1
B="python bench.py --steps 900 --warmup 50 --train-steps 0 --pixel-steps 0 --long-video-windows 0 --no-cpu"
for rep in 1 2; do
  timeout -k 10 200 $B 2>/dev/null > $OUT/b_fused.json; python -c "import sys,json; d=json.loads(open('$OUT/b_fused.json').read().strip().splitlines()[-1]); print('fused', d['value'], d['breakdown']['launches'])"
  LFVDM_SPATIAL_FUSED=0 timeout -k 10 200 $B 2>/dev/null > $OUT/b_unfused.json; python -c "import sys,json; d=json.loads(open('$OUT/b_unfused.json').read().strip().splitlines()[-1]); print('unfused', d['value'], d['breakdown']['launches'])"
done
python -c "
import json
d=json.loads(open('$OUT/b_fused.json').read().strip().splitlines()[-1])
for k,v in d['breakdown']['kernels'].items(): print(k, v)
"
