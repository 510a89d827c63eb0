#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c36
timeout 900 python -m pytest tests/test_midas_gpu.py tests/test_plugins_gpu.py -x -q -m gpu > gpurun_out/c36/tests.txt 2>&1; tail -3 gpurun_out/c36/tests.txt
for i in 1 2; do
for m in 0 1; do
CD_AMD_MIDAS_GRAD_IN_PLACE=$m timeout 600 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --frames 20 --steps 10 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/c36/midas_${m}_$i.json 2>gpurun_out/c36/midas_${m}_$i.err; echo "grad in place=$m $(cut -c90-200 gpurun_out/c36/midas_${m}_$i.json) $(python -c "
import json; d=json.loads(open('gpurun_out/c36/midas_${m}_$i.json').read().strip().splitlines()[-1]); print(d['config'].get('last_loss'))")"
done; done
