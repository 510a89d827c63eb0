#!/bin/bash
mkdir -p gpurun_out/r3
timeout 300 python tools/exp/diag_midas_nan.py 2>&1 | grep -v Warn | cut -c1-260 > gpurun_out/r3/diag_midas_nan.txt
grep -c "loss=nan" gpurun_out/r3/diag_midas_nan.txt; grep "^step" gpurun_out/r3/diag_midas_nan.txt | awk 'NR%4==1' | cut -c1-200
timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --no-cpu-baseline --no-config5 --no-loss-microbench --steps 20 --warmup 3 > gpurun_out/r3/bench_midas_f.json 2> gpurun_out/r3/bench_midas_f.log
tail -4 gpurun_out/r3/bench_midas_f.log; cut -c1-400 gpurun_out/r3/bench_midas_f.json
timeout 300 python -m pytest tests/test_midas_gpu.py -x -q -m gpu 2>&1 | tail -5
