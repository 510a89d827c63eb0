#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c23
echo "== staged fp32 kernels (CD_AMD_CONV1X1_KC=0)"; CD_AMD_CONV1X1_KC=0 timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c23/bench_fp32.txt
echo "== round 6"; timeout 300 python tools/exp/conv1x1_wide_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c23/bench_kc.txt
