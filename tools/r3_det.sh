#!/bin/bash
# round-3 development step: deterministic-gradient mode tests + its cost on the cfg-C training step
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
export LFVDM_TUNE_CACHE=$OUT/tune_new.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_new.json
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_backward_gpu.py tests/test_dist_gpu.py -m gpu -q -x --timeout 600 > $OUT/t_det.log 2>&1; rc=$?
tail -6 $OUT/t_det.log
[ $rc -ne 0 ] && exit 1
for det in 0 1; do
  LFVDM_DETERMINISTIC=$det timeout -k 10 300 python bench.py --steps 100 --train-steps 40 --no-cpu --pixel-steps 0 --long-video-windows 0 --no-breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deterministic=$det train ms/step', d['train']['ms_per_step'], 'loss', d['train']['last_loss'])"
done
