#!/bin/bash
# timing experiments on the sweep kernel (variants built by tools/exp/build_variants.sh with -DCD_SWEEP_EXP=n: 1 no source pass,
# 2 no flush, 4 no staging by the sources, 8 one item only; results are wrong by construction)
cd $GRAFT_REPO_ROOT
echo "== product"
python tools/loss_bench.py --batches 256,512,1024 --iters 40 --warm 100 --brief 2>&1 | tail -3
for n in 1 2 3 8; do
  echo "== exp$n"
  CD_AMD_LIB=tools/exp/variants/libcd_amd_exp$n.so python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2
done
