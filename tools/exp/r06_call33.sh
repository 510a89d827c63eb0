#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c33
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_midas_gpu.py -x -q -m gpu > gpurun_out/c33/tests.txt 2>&1; tail -3 gpurun_out/c33/tests.txt
for i in 1 2; do
timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c33/midas_$i.json 2>gpurun_out/c33/midas_$i.err; cut -c90-220 gpurun_out/c33/midas_$i.json
done
timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline --no-loss-microbench 2>/dev/null | cut -c90-200
