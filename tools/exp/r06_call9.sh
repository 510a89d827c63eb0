#!/bin/bash
# round 6, call 9: weight-gradient unpack with 16 loads in flight (product) vs 4 (unp4); wgrad tests (bit identity); loop test (burn-in fingerprint)
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
( timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_hourglass_engine_gpu.py "tests/test_loop_gpu.py::test_full_length_run_vs_fp64_and_vs_the_reference_fp32_run" -m gpu -q -x 2>&1 | tail -4 ) | tee gpurun_out/conv_tests_r06c9.txt
grep -h "burn_in_state_bitwise" gpurun_out/parity_log.txt | tail -3 | cut -c1-200
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
for rep in 1 2 3; do for v in base unp4; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/unp_variants.txt
