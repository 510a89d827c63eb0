#!/bin/bash
# round 6, call 17: unit samples with ONE memory latency (product) vs round 6's earlier build (units2lat): loss tests, the loss call A/B
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_loss_gpu.py -m gpu -q -x 2>&1 | tail -3 )
for rep in 1 2 3; do for v in base units2lat; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  echo "== $v"; CD_AMD_LIB=$L python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2
done; done | tee gpurun_out/units_variants.txt
grep -h "PARITY" gpurun_out/parity_log.txt | grep "roofline_launch\|baseline_size\[scene,384x224,v4" | cut -c1-220
