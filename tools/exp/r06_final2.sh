#!/bin/bash
# round 6, final record on a fresh box: the full GPU suite + smoke + the default bench line with the final build
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_20ep_r06b.txt
rm -f gpurun_out/parity_log.txt $CD_AMD_PARITY_CURVES
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r06.txt 2>&1
tail -n 8 gpurun_out/gpu_suite_r06.txt
unset CD_AMD_PARITY_CURVES
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r06_n1.json 2> gpurun_out/bench_r06_n1.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained']['frac'], d['roofline']['traffic_over_algorithmic'], d['roofline_conv']['frac'], d['cpu_baseline']['value'], d.get('config5',{}).get('value'), d['config']['host_ms_per_step'])
PY
python bench.py --frames 1000 --no-cpu-baseline --no-config5 --no-loss-microbench > gpurun_out/bench_r06_config3_n1.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_r06_config3_n1.json').read().strip().splitlines()[-1]); print('configs[3] clip N=1', d['value'], d['config']['host_ms_per_step'])"
