#!/bin/bash
# A/B of one variant library against the product, alternating (box-to-box differences are ~2 %): tools/exp/r05_ab.sh <variant name>
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== product"; python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2
  echo "== $1"; CD_AMD_LIB=tools/exp/variants/libcd_amd_$1.so python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2
done
CD_AMD_LIB=tools/exp/variants/libcd_amd_$1.so python -m pytest tests/test_loss_gpu.py -m gpu -x -q 2>&1 | tail -3
