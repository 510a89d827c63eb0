#!/bin/bash
echo "== base lib"; CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_base.so timeout 120 python tools/exp/bn_bench.py 2>&1 | grep -v amdgpu.ids
for p in 16 32 64 128; do echo "== new per_thread $p"; CD_AMD_BN_PER_THREAD=$p timeout 120 python tools/exp/bn_bench.py 2>&1 | grep -v amdgpu.ids | tail -12; done
timeout 300 python -m pytest tests/test_layers_gpu.py -x -q -m gpu 2>&1 | tail -2
