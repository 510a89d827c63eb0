#!/bin/bash
# round 6, call 22: chunked split-bf16 1x1 for wide filters + wide 1x1 wgrad: tests, config5 A/B, midas profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c22
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pointwise" > gpurun_out/c22/conv_test.txt 2>&1; tail -15 gpurun_out/c22/conv_test.txt
timeout 900 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "wgrad" > gpurun_out/c22/wgrad_test.txt 2>&1; tail -5 gpurun_out/c22/wgrad_test.txt
for i in 1 2; do
CD_AMD_CONV1X1_KC=0 timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c22/midas_base_$i.json 2>gpurun_out/c22/midas_base_$i.err; cut -c1-220 gpurun_out/c22/midas_base_$i.json
timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c22/midas_new_$i.json 2>gpurun_out/c22/midas_new_$i.err; cut -c1-220 gpurun_out/c22/midas_new_$i.json
done
bash tools/exp/prof_midas.sh hip > gpurun_out/c22/prof_midas.txt 2>&1; head -40 gpurun_out/c22/prof_midas.txt | cut -c1-180
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
