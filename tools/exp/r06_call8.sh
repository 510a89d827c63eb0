#!/bin/bash
# round 6, call 8: conv_split with the scale/shift table in LDS (product) vs + three resident workgroups for the 4-accumulator classes (occ3);
# the purity test; conv tests
set -u
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_hourglass_engine_gpu.py tests/test_finetune_gpu.py::test_the_step_launches_no_framework_kernels tests/test_midas_gpu.py -m gpu -q -x 2>&1 | tail -4 ) | tee gpurun_out/conv_tests_r06c8.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
for rep in 1 2 3; do for v in base occ3; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/occ3_variants.txt
