#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c37
CD_AMD_CONV1X1_KC_NW12=1 timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "wide_filters" > gpurun_out/c37/conv_test.txt 2>&1; tail -3 gpurun_out/c37/conv_test.txt
echo "== NW=8"; timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep "1024->1024\|512-> 512\|2048->2048" | tee gpurun_out/c37/bench_nw8.txt
echo "== NW=12 where it saves a round"; CD_AMD_CONV1X1_KC_NW12=1 timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep "1024->1024\|512-> 512\|2048->2048" | tee gpurun_out/c37/bench_nw12.txt
for m in 0 1 0 1; do
CD_AMD_CONV1X1_KC_NW12=$m timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench 2>/dev/null | cut -c90-200 | sed "s/^/nw12=$m /"
done
