#!/bin/bash
# round 6: the record of the FINAL tree: full GPU suite + smoke + the default bench line + the configs[3] clip
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_20ep_r06e.txt
rm -f gpurun_out/parity_log.txt $CD_AMD_PARITY_CURVES
( time timeout 2700 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r06e.txt 2>&1
tail -n 6 gpurun_out/gpu_suite_r06e.txt | cut -c1-300
unset CD_AMD_PARITY_CURVES
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r06e_n1.json 2> gpurun_out/bench_r06e_n1.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06e_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained']['frac'], d['roofline_conv']['frac'], d['cpu_baseline']['value'], d.get('config5',{}).get('value'))
PY
python bench.py --frames 1000 --no-cpu-baseline --no-config5 --no-loss-microbench > gpurun_out/bench_r06e_config3_n1.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_r06e_config3_n1.json').read().strip().splitlines()[-1]); print('configs[3] clip N=1', d['value'], d['config']['host_ms_per_step'])"
