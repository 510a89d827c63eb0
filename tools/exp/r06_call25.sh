#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c25
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pointwise" > gpurun_out/c25/conv_test.txt 2>&1; tail -4 gpurun_out/c25/conv_test.txt
echo "== NT=4 only"; CD_AMD_CONV1X1_KC_NT5=0 timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c25/bench_nt4.txt
echo "== NT=5 where it saves a round"; timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c25/bench_nt5.txt
for i in 1 2; do
CD_AMD_CONV1X1_KC_NT5=0 timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c25/midas_nt4_$i.json 2>gpurun_out/c25/midas_nt4_$i.err; cut -c90-220 gpurun_out/c25/midas_nt4_$i.json
timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c25/midas_nt5_$i.json 2>gpurun_out/c25/midas_nt5_$i.err; cut -c90-220 gpurun_out/c25/midas_nt5_$i.json
done
