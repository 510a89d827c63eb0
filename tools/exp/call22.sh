#!/bin/bash
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in base new b2_mfmaonly; do
  if [ $v = new ]; then unset CD_AMD_LIB; else export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_$v.so; fi
  echo "== $v"; timeout 300 python tools/conv_sweep.py --iters 10 --heur-only > gpurun_out/r3/conv_heur5_$v.txt 2>&1; tail -1 gpurun_out/r3/conv_heur5_$v.txt
done
