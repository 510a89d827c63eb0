#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c35
timeout 900 python -m pytest tests/test_midas_gpu.py -x -q -m gpu > gpurun_out/c35/tests.txt 2>&1; tail -3 gpurun_out/c35/tests.txt
for i in 1 2; do
for m in 0 1; do
CD_AMD_MIDAS_CONV_TUNE=$m timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c35/midas_${m}_$i.json 2>gpurun_out/c35/midas_${m}_$i.err; echo "conv tune=$m $(cut -c90-200 gpurun_out/c35/midas_${m}_$i.json)"
done; done
