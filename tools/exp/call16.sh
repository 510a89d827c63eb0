#!/bin/bash
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "weight_gradient or launch_shapes" 2>&1 | tail -2
for v in pf new t2 t14; do
  if [ $v = new ]; then unset CD_AMD_LIB; else export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_$v.so; fi
  echo "== $v"; timeout 300 python tools/wgrad_sweep.py --iters 10 > gpurun_out/r3/wgrad_sweep_$v.txt 2>&1; tail -1 gpurun_out/r3/wgrad_sweep_$v.txt
done
