#!/bin/bash
# round 6, call 31: is the run-to-run spread of the step (190 vs 196 pairs/s on one box) the launch-shape tuner's? Six runs with their own
# tune caches, then the best and the worst cache replayed three times each (no tuning in those runs).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c31
B="--steps 40 --warmup 10 --no-config5 --no-cpu-baseline --no-loss-microbench"
for i in 1 2 3 4 5 6; do
CD_AMD_CONV_TUNE_CACHE=$PWD/gpurun_out/c31/tune_$i.json timeout 300 python bench.py $B > gpurun_out/c31/bench_tune_$i.json 2>gpurun_out/c31/bench_tune_$i.err
done
python - <<'PY'
import json, shutil
v = {i: json.loads(open(f'gpurun_out/c31/bench_tune_{i}.json').read().strip().splitlines()[-1])['value'] for i in range(1, 7)}
print('tuning runs:', v)
best, worst = max(v, key=v.get), min(v, key=v.get)
shutil.copy(f'gpurun_out/c31/tune_{best}.json', 'gpurun_out/c31/tune_best.json'); shutil.copy(f'gpurun_out/c31/tune_{worst}.json', 'gpurun_out/c31/tune_worst.json')
print('best', best, 'worst', worst)
PY
for w in best worst best worst best worst; do
cp gpurun_out/c31/tune_$w.json /tmp/t.json
CD_AMD_CONV_TUNE_CACHE=/tmp/t.json timeout 300 python bench.py $B 2>/dev/null | python -c "import json,sys; print('replay $w', json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
done
