#!/bin/bash
# round 6, call 19: column-walking upsample forward: bits, microbench, serial step profile, A/B bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c19
timeout 600 python -m pytest tests/test_layers_gpu.py -x -q -m gpu > gpurun_out/c19/layers_test.txt 2>&1; tail -5 gpurun_out/c19/layers_test.txt
timeout 300 python tools/exp/layers_stream_bench.py > gpurun_out/c19/layers_bench.txt 2>&1; cat gpurun_out/c19/layers_bench.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_c19 $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_c19 --last-steps 4 > gpurun_out/prof_serial_c19/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_c19/summary4.txt > gpurun_out/c19/step_breakdown_serial.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_c19 --last-steps 4 --by-grid > gpurun_out/c19/step_kernels_by_grid.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
head -30 gpurun_out/c19/step_breakdown_serial.txt
for i in 1 2; do
CD_AMD_LAYERS_MODE=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline > gpurun_out/c19/bench_base_$i.json 2>gpurun_out/c19/bench_base_$i.err; cut -c1-200 gpurun_out/c19/bench_base_$i.json
timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline > gpurun_out/c19/bench_new_$i.json 2>gpurun_out/c19/bench_new_$i.err; cut -c1-200 gpurun_out/c19/bench_new_$i.json
done
