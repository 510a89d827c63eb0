#!/bin/bash
# round 6, call 20: no alias fan-in in the python engine: engine tests (bitwise vs the C engine), A/B bench, serial profile
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c20
timeout 900 python -m pytest tests/test_hourglass_engine_gpu.py tests/test_finetune_gpu.py -x -q -m gpu > gpurun_out/c20/engine_test.txt 2>&1; tail -5 gpurun_out/c20/engine_test.txt
cp consistent_depth_amd/monodepth/hourglass_engine.py /tmp/new_engine.py
for i in 1 2; do
cp tools/exp/variants/hourglass_engine_alias.py consistent_depth_amd/monodepth/hourglass_engine.py
timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline > gpurun_out/c20/bench_alias_$i.json 2>gpurun_out/c20/bench_alias_$i.err; cut -c1-200 gpurun_out/c20/bench_alias_$i.json
cp /tmp/new_engine.py consistent_depth_amd/monodepth/hourglass_engine.py
timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline > gpurun_out/c20/bench_new_$i.json 2>gpurun_out/c20/bench_new_$i.err; cut -c1-200 gpurun_out/c20/bench_new_$i.json
done
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_c20 $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_c20 --last-steps 4 > gpurun_out/prof_serial_c20/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_c20/summary4.txt > gpurun_out/c20/step_breakdown_serial.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_c20 --last-steps 4 --by-grid > gpurun_out/c20/step_kernels_by_grid.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
head -8 gpurun_out/c20/step_breakdown_serial.txt; grep add_slice gpurun_out/c20/step_breakdown_serial.txt
