#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c24
OUT=$PWD/gpurun_out/prof_midas_c24; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
CD_AMD_MIDAS_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-loss-microbench > $OUT/trace.log 2>&1
cd $REPO
python tools/prof_step_summary.py $OUT --last-steps 4 --by-grid > gpurun_out/c24/midas_by_grid.txt 2>&1
find $OUT -name "*.db" -delete; rm -rf $OUT/trace
head -60 gpurun_out/c24/midas_by_grid.txt | cut -c1-210
