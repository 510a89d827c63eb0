#!/bin/bash
# round 6: bench line + rocprofv3 evidence of the FINAL tree from ONE box (the suite record of the same tree: r06_final8.sh)
set -u
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r06f_n1.json 2> gpurun_out/bench_r06f_n1.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06f_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], d['roofline']['sustained']['frac'], d['roofline_conv']['frac'], d['cpu_baseline']['value'], d.get('config5',{}).get('value'))
PY
bash tools/prof_loss.sh r06h --batches 256 --iters 200 --warm 100 > /dev/null 2>&1
head -4 gpurun_out/prof_r06h/summary.txt
bash tools/prof_bench.sh r06f > gpurun_out/prof_bench_r06f.log 2>&1
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r06h $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06h --last-steps 4 > gpurun_out/prof_serial_r06h/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r06h/summary4.txt > gpurun_out/step_breakdown_serial_r06h.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06h --last-steps 4 --by-grid > gpurun_out/step_kernels_by_grid_r06h.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out -type d -name "pmc_*" -prune -exec rm -rf {} + 2>/dev/null
head -12 gpurun_out/step_breakdown_serial_r06h.txt
