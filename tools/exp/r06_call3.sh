#!/bin/bash
# round 6, call 3: loss tests with the workspace header (ABI 9); loss call: product vs non-temporal variants; hand-written streaming reference
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_loss_gpu.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -5 ) | tee gpurun_out/loss_tests_r06c3.txt
tools/hbm_stream 256 40 | tee gpurun_out/hbm_stream_256.txt
tools/hbm_stream 1024 20 | tee gpurun_out/hbm_stream_1024.txt
for rep in 1 2; do
for v in base nt1 nt2 nt3 nt7; do
  L=""; [ $v != base ] && L=tools/exp/variants/libcd_amd_$v.so
  echo "== $v" ; CD_AMD_LIB=$L python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -3
done; done | tee gpurun_out/loss_nt_variants.txt
