#!/bin/bash
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
export LFVDM_TUNE_CACHE=$OUT/tune_new.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_new.json
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py tests/test_train_gpu.py -m gpu -q -x --timeout 600 -k "temporal or backward or gradient or cfgC or deterministic or wgrad or conv" > $OUT/t_rows.log 2>&1; rc=$?
tail -4 $OUT/t_rows.log
[ $rc -ne 0 ] && exit 1
for v in 0 1; do
  if [ $v = 1 ]; then export LFVDM_ATTN_BWD_ROWS_V1=1; fi
  timeout -k 10 300 python bench.py --steps 100 --train-steps 40 --no-cpu --pixel-steps 0 --long-video-windows 0 --no-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows_v1=$v train ms/step', d['train']['ms_per_step'], 'frac', d['train']['roofline']['frac'])"
done
