#!/bin/bash
for f in smooth random; do
echo base; MASK_FLOW=$f CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_base.so timeout 120 python tools/exp/mask_bench.py 2>&1 | grep -v amdgpu.ids
echo new; MASK_FLOW=$f timeout 120 python tools/exp/mask_bench.py 2>&1 | grep -v amdgpu.ids
done
