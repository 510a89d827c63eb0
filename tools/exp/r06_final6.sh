#!/bin/bash
# round 6, final record after the streaming layers / fan-in / wide 1x1 work: GPU suite + smoke + default bench line, configs[3] clip, then rocprofv3 evidence
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_20ep_r06c.txt
rm -f gpurun_out/parity_log.txt $CD_AMD_PARITY_CURVES
( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r06c.txt 2>&1
tail -n 8 gpurun_out/gpu_suite_r06c.txt
unset CD_AMD_PARITY_CURVES
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r06c_n1.json 2> gpurun_out/bench_r06c_n1.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06c_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained']['frac'], d['roofline_conv']['frac'], d['cpu_baseline']['value'], d.get('config5',{}).get('value'), d.get('config5',{}).get('roofline_conv',{}).get('frac'))
PY
python bench.py --frames 1000 --no-cpu-baseline --no-config5 --no-loss-microbench > gpurun_out/bench_r06c_config3_n1.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_r06c_config3_n1.json').read().strip().splitlines()[-1]); print('configs[3] clip N=1', d['value'], d['config']['host_ms_per_step'])"
bash tools/prof_loss.sh r06g --batches 256 --iters 200 --warm 100 > /dev/null 2>&1
head -14 gpurun_out/prof_r06g/summary.txt
bash tools/prof_bench.sh r06c > gpurun_out/prof_bench_r06c.log 2>&1
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r06g $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06g --last-steps 4 > gpurun_out/prof_serial_r06g/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r06g/summary4.txt > gpurun_out/step_breakdown_serial_r06g.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06g --last-steps 4 --by-grid > gpurun_out/step_kernels_by_grid_r06g.txt 2>&1
bash tools/exp/prof_midas.sh hip > gpurun_out/prof_midas_r06g.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out -type d -name "pmc_*" -prune -exec rm -rf {} + 2>/dev/null
head -14 gpurun_out/step_breakdown_serial_r06g.txt
