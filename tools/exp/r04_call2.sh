#!/bin/bash
set -u
OUT=gpurun_out/r04_2; mkdir -p $OUT
export CD_AMD_CONV_TUNE_CACHE=$PWD/$OUT/conv_tune.json
timeout 600 python -m pytest tests/test_conv_gpu.py -q -x -k "many_convolutions or one_dispatch" > $OUT/t_conv.log 2>&1
tail -n 3 $OUT/t_conv.log
timeout 900 python -m pytest tests/test_hourglass_engine_gpu.py -q -x -k "2x64x96 or handle" > $OUT/t_engine.log 2>&1
tail -n 5 $OUT/t_engine.log
for b in 1 0 1 0; do export CD_AMD_CONV_MULTI=$b;
  timeout 200 python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch=$b', d['value'], d['ms_per_step'])"
done
