#!/bin/bash
# rocprofv3 kernel trace of the full fine-tuning step.  usage: bash tools/prof_step.sh <tag> [bench args...]
set -u
TAG=${1:-step}; shift || true
ARGS=${@:---backend hip --steps 3 --warmup 2 --no-cpu-baseline --no-config5 --no-loss-microbench}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
cd $REPO
python tools/prof_step_summary.py $OUT | tee $OUT/summary.txt
