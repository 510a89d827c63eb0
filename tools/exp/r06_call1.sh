#!/bin/bash
# round 6, call 1: the burn-in snapshot of clip "a" (stage 1 of oracle/gen_golden_loop_384.py) for the direct product-vs-reference-fp32
# comparison, the default bench line on this round's first box and the serial step breakdown
set -u
cd $GRAFT_REPO_ROOT
( time python -m oracle.gen_golden_loop_384 snapshot a gpurun_out/snap384 ) > gpurun_out/snap_a.log 2>&1
tail -n 4 gpurun_out/snap_a.log
python bench.py --no-config5 > gpurun_out/bench_r06_c1.json 2> gpurun_out/bench_r06_c1.log
tail -c 1800 gpurun_out/bench_r06_c1.json
bash tools/prof_step.sh r06c1 > /dev/null 2>&1
head -60 gpurun_out/prof_r06c1/summary.txt
find gpurun_out/prof_r06c1 -name "*.db" -delete
du -sh gpurun_out/*
