#!/bin/bash
# SQ / traffic counter passes over tools/profile_kernels.py (separate --pmc passes, no trace domains).
# usage (on the GPU box, from the repo root): bash tools/gpu_pmc.sh <tag> [batch] [reps] [n w k d]
set -e
TAG=$1; B=${2:-4096}; R=${3:-10}; SHAPE="${4:-} ${5:-} ${6:-} ${7:-}"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS \
  --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/profile_kernels.py $B $R $SHAPE > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/profile_kernels.py $B $R $SHAPE > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/profile_kernels.py $B $R $SHAPE > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 $ROOT/tools/profile_kernels.py $B $R $SHAPE > $OUT/p4.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
