#!/bin/bash
# round 6, call 7: forward / input-gradient kernel with the scale/shift table in LDS (pf0) and the first-unit prefetch (product: MG 2 in the
# 128-accumulator classes; pf1mg4: MG 4): correctness (bit-identity tests), serial family times, end-to-end bench, alternating
set -u
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_hourglass_engine_gpu.py tests/test_layers_gpu.py -m gpu -q -x 2>&1 | tail -4 ) | tee gpurun_out/conv_tests_r06c7.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
for v in pf0 base pf1mg4; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh pf_$v $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
  python tools/prof_step_summary.py gpurun_out/prof_pf_$v --last-steps 4 > gpurun_out/prof_pf_$v/summary4.txt 2>&1
  python tools/prof_families.py gpurun_out/prof_pf_$v/summary4.txt > gpurun_out/pf_families_$v.txt 2>&1
  python tools/prof_step_summary.py gpurun_out/prof_pf_$v --last-steps 4 --by-grid > gpurun_out/pf_bygrid_$v.txt 2>&1
  echo "== $v"; grep "conv_fwd_split\|^sum" gpurun_out/pf_families_$v.txt
done
for rep in 1 2; do for v in pf0 base pf1mg4; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/pf_variants.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
