#!/bin/bash
# (the CD_AMD_WGRAD_MERGE switch of this A/B -- the three weight gradients of an inception in one dispatch -- was measured slower and is not in the tree:
#  profiles/wgrad_batch_r04.txt; kept as the record of the command line)
set -u
export CD_AMD_CONV_TUNE_CACHE=$PWD/gpurun_out/conv_tune.json
CD_AMD_WGRAD_MERGE=1 timeout 600 python -m pytest tests/test_hourglass_engine_gpu.py -q -x -k "2x64x96 or handle" 2>&1 | tail -2
for b in 1 0 1 0; do
  CD_AMD_WGRAD_MERGE=$b timeout 200 python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_merge=$b', d['value'], d['ms_per_step'])"
done
