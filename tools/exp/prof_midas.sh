#!/bin/bash
# kernel trace of the configs[4] side measurement with the dense 1x1 convolutions on the hand-written kernels (hip) or on the library (gemm)
set -u
MODE=${1:-hip}
OUT=$PWD/gpurun_out/prof_midas_$MODE
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CD_AMD_MIDAS_1X1=$MODE CD_AMD_MIDAS_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-loss-microbench > $OUT/trace.log 2>&1
cd $REPO
python tools/prof_step_summary.py $OUT --last-steps 4 > $OUT/summary_trace.txt 2>&1
head -45 $OUT/summary_trace.txt
find $OUT -name "*.db" -delete
