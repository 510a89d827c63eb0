#!/bin/bash
# round 6: rocprofv3 evidence of the FINAL tree from one box: bench.py's own line, then the loss call's kernel trace + PMC, the bench's kernel trace + PMC, the serial breakdown
set -u
cd $GRAFT_REPO_ROOT
python bench.py --no-config5 > gpurun_out/bench_r06_profiled_box.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_r06_profiled_box.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['frac'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['sustained']['frac'])"
bash tools/prof_loss.sh r06f --batches 256 --iters 200 --warm 100 > /dev/null 2>&1
head -14 gpurun_out/prof_r06f/summary.txt
bash tools/prof_bench.sh r06b > gpurun_out/prof_bench_r06b.log 2>&1
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r06f $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06f --last-steps 4 > gpurun_out/prof_serial_r06f/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r06f/summary4.txt > gpurun_out/step_breakdown_serial_r06f.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06f --last-steps 4 --by-grid > gpurun_out/step_kernels_by_grid_r06f.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out -type d -name "pmc_*" -prune -exec rm -rf {} + 2>/dev/null
head -14 gpurun_out/step_breakdown_serial_r06f.txt
