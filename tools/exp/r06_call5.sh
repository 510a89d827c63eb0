#!/bin/bash
# round 6, call 5: full GPU suite; the forward kernel's staging split into its memory round trip and its arithmetic (CD_SP_DBG builds,
# serial eager traces); which non-cd:: kernels still run inside a step
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
( time timeout 1800 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/gpu_suite_r06b.txt 2>&1
tail -n 12 gpurun_out/gpu_suite_r06b.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench --steps 4 --warmup 3 --graph 0"
for v in base spdbg4 spdbg2; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh sp_$v $B > /dev/null 2>&1
  python tools/prof_step_summary.py gpurun_out/prof_sp_$v --last-steps 4 > gpurun_out/prof_sp_$v/summary4.txt 2>&1
  python tools/prof_families.py gpurun_out/prof_sp_$v/summary4.txt > gpurun_out/sp_families_$v.txt 2>&1
  python tools/prof_step_summary.py gpurun_out/prof_sp_$v --last-steps 4 --by-grid > gpurun_out/sp_bygrid_$v.txt 2>&1
  echo "== $v"; grep "conv_fwd_split\|^sum" gpurun_out/sp_families_$v.txt
done
python tools/prof_aten.py gpurun_out/prof_sp_base 4 | tee gpurun_out/aten_in_step_r06.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
