#!/bin/bash
# round 6, call 10: weight-gradient unpack: 16 loads in flight (product) vs 32 (u32) vs 16 + 128 workgroups per descriptor (g128)
set -u
cd $GRAFT_REPO_ROOT
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
for rep in 1 2 3; do for v in base u32 g128; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/unp_variants2.txt
