#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c32
timeout 600 python -m pytest tests/test_layers_gpu.py -x -q -m gpu -k "streaming or pool" > gpurun_out/c32/layers_test.txt 2>&1; tail -3 gpurun_out/c32/layers_test.txt
timeout 300 python tools/exp/layers_stream_bench.py 2>&1 | grep "upsample2x_add_fwd\|sum of" | tee gpurun_out/c32/layers_bench.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_c32 $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_c32 --last-steps 4 --by-grid > gpurun_out/c32/step_kernels_by_grid.txt 2>&1
find gpurun_out -name "*.db" -delete; find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
head -1 gpurun_out/c32/step_kernels_by_grid.txt; grep "upsample2x_add_fwd" gpurun_out/c32/step_kernels_by_grid.txt | cut -c1-180
