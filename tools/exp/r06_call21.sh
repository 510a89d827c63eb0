#!/bin/bash
# round 6, call 21: two alternating graph executables: graph tests, A/B bench (CD_AMD_GRAPH_EXECS=1 vs 2 vs 3), gaps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c21
timeout 900 python -m pytest tests/test_finetune_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/c21/graph_tests.txt 2>&1; tail -4 gpurun_out/c21/graph_tests.txt
for i in 1 2; do
for n in 1 2 3; do
CD_AMD_GRAPH_EXECS=$n timeout 300 python bench.py --steps 60 --warmup 10 --no-config5 --no-cpu-baseline --no-loss-microbench > gpurun_out/c21/bench_execs${n}_$i.json 2>gpurun_out/c21/bench_execs${n}_$i.err; echo "execs=$n $(cut -c1-200 gpurun_out/c21/bench_execs${n}_$i.json)"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c21/bench_execs*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d.get('host_ms_per_step'), d.get('host_loop_ms_per_step'))
PY
