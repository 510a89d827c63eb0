#!/bin/bash
for m in level both branch level both; do
  CD_AMD_ENGINE_STREAMS=$m timeout 300 python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['config'].get('hip_graph'))"
done
CD_AMD_ENGINE_WGRAD_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('level+wgradstream', d['value'], d['ms_per_step'], d['config'].get('hip_graph'))"
