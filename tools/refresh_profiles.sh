#!/bin/bash
# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun; results land in
# gpurun_out/refresh/ and are copied into profiles/ afterwards, prefixed with the round: ROUND=r03 by default):
#   bash tools/refresh_profiles.sh
# 1. rocprofv3 kernel stats, training step                 -> train_kernel_stats.csv (also placed in profiles/ on the
#    box, so that the bench line's train.roofline.families is computed from THIS profile)
# 2. (moved: the default bench line is taken after the PMC passes, step 4b)
# 3. rocprofv3 kernel stats, sampling bench                -> bench_kernel_stats.csv
# 4. PMC passes over eager denoising steps (tools/pmc_target.py), one counter group per pass:
#    FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE  -> pmc_traffic.json (+ raw counter CSVs)
# A step that is killed at its limit stops the chain.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh
ROUND=${ROUND:-r06}
# STAGES (default: all) selects what runs - one gpurun call is limited to 20 minutes, the whole refresh takes longer:
#   STAGES="train pixel" | "bench pmc line" | "stamps parity" | "long" | "localbench"      (results accumulate in gpurun_out/refresh/)
STAGES=${STAGES:-"train pixel bench pmc line stamps parity long localbench"}
want() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
[ "${KEEP_OUT:-0}" = "1" ] || rm -rf $OUT
mkdir -p $OUT
# tune codes measured by earlier stages of this refresh are read back (second cache file) and extended
export LFVDM_TUNE_CACHE=$ROOT/profiles/tune_cache_mi355x.json      # read-only
export LFVDM_TUNE_CACHE_OUT=$OUT/tune_cache_mi355x.json             # committed table + anything measured in these runs
cd /tmp && export TMPDIR=/tmp
step() { local lim=$1; shift; timeout -k 10 $lim "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi; return 0; }
# Every stage stamps ITS OWN result with the code it ran (library ABI + digest of csrc/* and the headers) the moment the result
# exists, and FAILS if that does not work: a profile without the stamp of its own run certifies nothing (round 5: a shell that
# did not know `stamp` left two profiles carrying a copy of another run's stamp).  stamp <file> [<name under profiles/>]
stamp() {
python3 - "$1" <<PY || { echo "stamp FAILED for $1: stopping"; exit 1; }
import json, os, sys, time
sys.path.insert(0, "$ROOT"); sys.path.insert(0, os.path.join("$ROOT", "latent-flexible-video-diffusion-modeling_amd"))
import bench
assert os.path.getsize(sys.argv[1]) > 0, "empty result"
st = bench.running_code_stamp()
st["git_head"] = os.environ.get("GIT_HEAD") or None
st["result"] = os.path.basename(sys.argv[1])
st["result_mtime"] = os.path.getmtime(sys.argv[1])
st["stamped_at"] = time.time()
json.dump(st, open(sys.argv[1] + ".stamp.json", "w"))
PY
[ -s "$1.stamp.json" ] || { echo "no stamp written for $1: stopping"; exit 1; }
if [ -n "$2" ]; then cp "$1" "$ROOT/profiles/${ROUND}_$2" && cp "$1.stamp.json" "$ROOT/profiles/${ROUND}_$2.stamp.json" || exit 1; fi
}
# (launch shapes that are not in the committed table yet are measured in an unprofiled pass first: both cache files are read)
if want train; then
step 300 python3 $ROOT/tools/train_profile.py 4 > $OUT/tpwarm.log 2>&1
step 300 rocprofv3 --kernel-trace --stats -d $OUT/tp -o tp --output-format csv -- python3 $ROOT/tools/train_profile.py 30 > $OUT/tp.log 2>&1
cp $OUT/tp/tp_kernel_stats.csv $OUT/train_kernel_stats.csv
stamp $OUT/train_kernel_stats.csv train_kernel_stats.csv
echo "train profile done"
fi
if want pixel; then
# 1b. pixel-space training (README recipe at batch 1): tune first (outside the profile), then profile 6 steps
step 300 python3 $ROOT/tools/pixel_train_profile.py --batch 1 --rb 1 --steps 2 > $OUT/pxwarm.log 2>&1
step 300 rocprofv3 --kernel-trace --stats -d $OUT/pxp -o pxp --output-format csv -- python3 $ROOT/tools/pixel_train_profile.py --batch 1 --rb 1 --steps 6 --warmup 4 > $OUT/pxp.log 2>&1
cp $OUT/pxp/pxp_kernel_stats.csv $OUT/pixel_train_kernel_stats.csv
stamp $OUT/pixel_train_kernel_stats.csv pixel_train_kernel_stats.csv
rm -rf $OUT/pxp/*trace*
echo "pixel train profile done"
fi
if want bench; then
step 300 python3 $ROOT/bench.py --steps 100 --warmup 20 --train-steps 0 --pixel-steps 0 --long-video-windows 4 --no-cpu --no-breakdown > $OUT/bpwarm.log 2>&1   # (tunes what the table lacks)
step 300 rocprofv3 --kernel-trace --stats -d $OUT/bp -o bp --output-format csv -- python3 $ROOT/bench.py --steps 300 --warmup 20 --train-steps 0 --pixel-steps 0 --long-video-windows 0 --no-cpu > $OUT/bp.log 2>&1
cp $OUT/bp/bp_kernel_stats.csv $OUT/bench_kernel_stats.csv
stamp $OUT/bench_kernel_stats.csv bench_kernel_stats.csv
echo "bench profile done"
fi
if want pmc; then
step 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_target.py > $OUT/pmc_fetch.log 2>&1
echo "pmc fetch done"
step 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_target.py > $OUT/pmc_write.log 2>&1
echo "pmc write done"
step 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ROOT/tools/pmc_target.py > $OUT/pmc_mfma.log 2>&1
echo "pmc mfma done"
python3 $ROOT/tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json $OUT/pmc_mfma
stamp $OUT/pmc_traffic.json pmc_traffic.json
fi
if want line; then
# 4b. the default bench line LAST among the measurements it quotes: train / pixel-train family splits and PMC traffic are
#     read from profiles/${ROUND}_*, which must be THIS run's files (stamped with the running code)
step 500 python3 $ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench done"
fi
# 5. in-kernel phase table of the implicit-GEMM launches (diagnostic build with -DLFVDM_STAMP, if present)
if want stamps && [ -f $ROOT/devlib/liblfvdm_chainstamp.so ]; then
  cd $ROOT && LFVDM_TUNE_CACHE_OUT= step 200 python3 tools/chain_stamps.py > $OUT/chain_stamps.txt 2>&1; cd /tmp
  echo "chain stamps done"
fi
if want stamps && [ -f $ROOT/devlib/liblfvdm_stamp.so ]; then
  cd $ROOT && LFVDM_TUNE_CACHE_OUT= step 200 python3 tools/conv_phase_stamps.py all > $OUT/conv_phase_stamps.txt 2>&1; cd /tmp
  echo "phase stamps done"
fi
# 6. the parity tests that print their deviations from the reference fixtures
if want parity; then
cd $ROOT && step 1100 python3 -m pytest tests/test_forward_gpu.py tests/test_backward_gpu.py tests/test_sampler_gpu.py tests/test_train_gpu.py -m gpu -q -s -k "reference or cfgC_training or replayed or full_size or fp64 or parameter_gradients or drift" --timeout 1200 > $OUT/parity_deviations.txt 2>&1; cd /tmp
echo "parity deviations done"
fi
if want long; then
# 7. the whole 1000-frame hierarchy-2 video (BASELINE.json configs[3] at full size: 97 windows x 250 steps)
step 400 python3 $ROOT/bench.py --steps 50 --warmup 10 --train-steps 0 --pixel-steps 0 --no-cpu --long-video-windows 97 > $OUT/long_video_line.json 2> $OUT/long_video.err
echo "long video done"
fi
if want localbench; then
# 8. micro-benchmark of the sample-local chain stage against the split-K tile body (tools/local_stage_bench.py), its in-kernel
#    phase stamps (diagnostic build, if present) and the fabric-side bytes per stage of both bodies (one counter per pass)
cd $ROOT
step 200 python3 tools/local_stage_bench.py $OUT/local_stage_bench.json > $OUT/lsb.log 2>&1
stamp $OUT/local_stage_bench.json local_stage_bench.json
if [ -f $ROOT/devlib/liblfvdm_chainstamp.so ]; then
  LFVDM_LIB_PATH=$ROOT/devlib/liblfvdm_chainstamp.so LOCAL_BENCH_STAMPS=1 step 200 python3 tools/local_stage_bench.py $OUT/local_stage_bench_stamps.json > $OUT/lsbs.log 2>&1
fi
cd /tmp
for V in tile:2 local:2:1 tile:4 local:4:1 local:4:2; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    LOCAL_BENCH_ONLY=$V step 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/lsb_pmc/${V//:/_}_$CNT -- python3 $ROOT/tools/local_stage_bench.py > $OUT/lsb_pmc_${V//:/_}_$CNT.log 2>&1
  done
done
python3 $ROOT/tools/local_stage_pmc.py $OUT/lsb_pmc $OUT/local_stage_bench.json $OUT/local_stage_pmc.json
stamp $OUT/local_stage_pmc.json local_stage_pmc.json
echo "local stage bench done"
fi
rm -rf $OUT/bp/*trace* $OUT/tp/*trace*
[ -f $LFVDM_TUNE_CACHE_OUT ] || cp $LFVDM_TUNE_CACHE $LFVDM_TUNE_CACHE_OUT
echo "refresh complete"
