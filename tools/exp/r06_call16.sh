#!/bin/bash
# round 6, call 16: forward-only (validation) loss kernel with non-temporal flow / mask loads (product) vs default policy (v1nt0); its tests
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_loss_gpu.py -m gpu -q -x 2>&1 | tail -3 )
for rep in 1 2 3; do for v in base v1nt0; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  echo "== $v"; CD_AMD_LIB=$L python tools/loss_bench.py --fwd-only --batches 4,256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -3
done; done | tee gpurun_out/v1_nt_variants.txt
python bench.py --no-config5 > gpurun_out/bench_r06_box_c16.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/bench_r06_box_c16.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['frac'], d['roofline']['sustained']['frac'])"
