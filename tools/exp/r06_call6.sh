#!/bin/bash
# round 6, call 6: the tests touched by the framework-kernel removal and the direct parity column; bench line
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_curves_r06.txt
rm -f $CD_AMD_PARITY_CURVES
( time timeout 1800 python -m pytest tests/test_finetune_gpu.py tests/test_loop_gpu.py tests/test_hourglass_engine_gpu.py tests/test_dp_gpu.py tests/test_midas_gpu.py tests/test_driver_gpu.py tests/test_loss_gpu.py -m gpu -q --durations=5 ) > gpurun_out/gpu_suite_r06c.txt 2>&1
tail -n 14 gpurun_out/gpu_suite_r06c.txt
python bench.py --no-config5 > gpurun_out/bench_r06_c6.json 2> gpurun_out/bench_r06_c6.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r06_c6.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d['config']['host_loop_ms_per_step'], d['roofline']['frac'], d['roofline'].get('sustained'), d['roofline_in_step']['frac'])
PY
