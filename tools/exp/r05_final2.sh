#!/bin/bash
# the default bench line and the loss call's rocprofv3 evidence from ONE box (boxes differ: see profiles/bench_r05_boxes.txt)
set -u
cd $GRAFT_REPO_ROOT
python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2 > gpurun_out/loss_bench_r05c.txt
python bench.py > gpurun_out/bench_r05c_n1.json 2> gpurun_out/bench_r05c_n1.log
python tools/hbm_ref.py > gpurun_out/hbm_ref_r05c.json 2>/dev/null
bash tools/prof_loss.sh r05c --batches 256 --iters 40 --warm 100 > /dev/null 2>&1
find gpurun_out/prof_r05c -name "*.db" -delete
rocm-smi --showclocks --showpower --showtemp > gpurun_out/smi_r05c.txt 2>&1
cat gpurun_out/loss_bench_r05c.txt; head -3 gpurun_out/prof_r05c/summary.txt
python -c "
import json; d=json.loads(open('gpurun_out/bench_r05c_n1.json').read().strip().splitlines()[-1]); print(d['value'], {k:d['roofline'][k] for k in ('frac','avg_ms','traffic_over_algorithmic','sustained')})"
