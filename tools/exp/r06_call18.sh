#!/bin/bash
# round 6, call 18: the streaming layer kernels (bits vs the scalar ones, microbench), then the step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c18
timeout 600 python -m pytest tests/test_layers_gpu.py -x -q -m gpu > gpurun_out/c18/layers_test.txt 2>&1; tail -5 gpurun_out/c18/layers_test.txt
timeout 300 python tools/exp/layers_stream_bench.py > gpurun_out/c18/layers_bench.txt 2>&1; cat gpurun_out/c18/layers_bench.txt
timeout 600 python -m pytest tests/test_hourglass_engine_gpu.py -x -q -m gpu > gpurun_out/c18/engine_test.txt 2>&1; tail -3 gpurun_out/c18/engine_test.txt
for i in 1 2; do
CD_AMD_LAYERS_MODE=1 timeout 300 python bench.py --steps 40 --warmup 10 > gpurun_out/c18/bench_base_$i.json 2>gpurun_out/c18/bench_base_$i.err; cut -c1-200 gpurun_out/c18/bench_base_$i.json
timeout 300 python bench.py --steps 40 --warmup 10 > gpurun_out/c18/bench_new_$i.json 2>gpurun_out/c18/bench_new_$i.err; cut -c1-200 gpurun_out/c18/bench_new_$i.json
done
