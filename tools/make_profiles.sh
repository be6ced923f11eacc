#!/bin/bash
# Round-2 profile artefacts (run on the GPU box from the repo root; results land in gpurun_out/prof_r02/,
# tools/collect_profiles.py then files them under profiles/).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (a) kernel trace + stats of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --skip-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# (b) one launch size per file: fused / staged kernels, fp32 and bf16 storage
for B in 512 4096 32768; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kern_B$B -- python3 $ROOT/tools/profile_kernels.py $B 10 > $OUT/kern_B$B.log 2>&1
done
# (c) counters: separate --pmc passes, no trace domains
cd $ROOT
for B in 512 4096 32768; do bash tools/gpu_pmc.sh r02_B$B $B 6 > $OUT/pmc_B$B.log 2>&1; done
# (d) the un-profiled bench line
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
# (e) one training step (eager launches so that the trace shows every kernel)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 $ROOT/tools/profile_train.py 512 30 eager > $OUT/train.log 2>&1
echo profiles-done
