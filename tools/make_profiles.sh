#!/bin/bash
# Profile artefacts of a round (run on the GPU box from the repo root; results land in gpurun_out/prof_<tag>/,
# tools/collect_profiles.py <tag> then files them under profiles/).   usage: bash tools/make_profiles.sh [tag] [part]
# part: all (default) | bench | kernels | pmc | train | config4 — the whole set is ~8 GPU-minutes, in parts it
# fits several short gpurun calls.
set -e
TAG=${1:-r03}; PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $PART = all ] || [ $PART = bench ]; then
  # (a) kernel trace + stats of the bench command itself, then the un-profiled bench line
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --skip-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  (cd $ROOT && python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err)
fi
if [ $PART = all ] || [ $PART = kernels ]; then
  # (b) one launch size per file: fused / staged kernels, fp32 and bf16 storage
  for B in 512 4096 32768; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kern_B$B -- python3 $ROOT/tools/profile_kernels.py $B 30 > $OUT/kern_B$B.log 2>&1
  done
fi
if [ $PART = all ] || [ $PART = pmc ]; then
  # (c) counters: separate --pmc passes, no trace domains
  (cd $ROOT && for B in 512 4096 32768; do bash tools/gpu_pmc.sh ${TAG}_B$B $B 6 > $OUT/pmc_B$B.log 2>&1; done)
fi
if [ $PART = all ] || [ $PART = train ]; then
  # (e) one training step (eager launches so that the trace shows every kernel)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $ROOT/tools/profile_train.py 512 30 eager > $OUT/train.log 2>&1
fi
if [ $PART = all ] || [ $PART = config4 ]; then
  # (f) BASELINE configs[4] (512 sensors, top-k 64, W=30, d=64): the row-gather family, stats + counters at 4096 windows
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kern_config4_B4096 -- python3 $ROOT/tools/profile_kernels.py 4096 6 512 30 64 64 > $OUT/kern_config4.log 2>&1
  (cd $ROOT && bash tools/gpu_pmc.sh ${TAG}_config4_B4096 4096 4 512 30 64 64 > $OUT/pmc_config4.log 2>&1)
fi
echo profiles-done
