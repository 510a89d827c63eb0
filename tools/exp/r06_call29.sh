#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c29
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pointwise" > gpurun_out/c29/conv_test.txt 2>&1; tail -3 gpurun_out/c29/conv_test.txt
timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c29/bench_kc.txt
for i in 1 2; do
timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c29/midas_$i.json 2>gpurun_out/c29/midas_$i.err; cut -c90-220 gpurun_out/c29/midas_$i.json
done
