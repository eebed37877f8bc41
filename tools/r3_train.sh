#!/bin/bash
# round-3 development step: cfg-C step test, training bench + rocprofv3 kernel stats of the training step
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
ROOT=$PWD
export LFVDM_TUNE_CACHE=$OUT/tune_new.json LFVDM_TUNE_CACHE_OUT=$OUT/tune_new.json
timeout -k 10 300 python -m pytest tests/test_train_gpu.py -m gpu -q -x --timeout 300 -k "cfgC_training" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 200 --train-steps 40 --no-cpu --pixel-steps 0 --long-video-windows 0 --no-breakdown > $OUT/bench_train.json 2> $OUT/bench_train.err || exit 1
python -c "import json; d=json.loads(open('$OUT/bench_train.json').read().strip().splitlines()[-1]); print('sample', d['value'], 'train', d['train']['ms_per_step'], d['train']['roofline']['frac'], d['train']['host_issue_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/tp -o tp --output-format csv -- python3 $ROOT/tools/train_profile.py 30 > $OUT/tp.log 2>&1 || exit 1
cp $OUT/tp/tp_kernel_stats.csv $OUT/train_kernel_stats_a.csv
rm -rf $OUT/tp
tail -5 $OUT/tp.log
