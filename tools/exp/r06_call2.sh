#!/bin/bash
# round 6, call 2: dense snapshot; graph-replay gap analysis; serial per-family breakdown (start of round); priority variants of conv_split
set -u
cd $GRAFT_REPO_ROOT
( time python -m oracle.gen_golden_loop_384 snapshot dense gpurun_out/snap384 ) > gpurun_out/snap_dense.log 2>&1
tail -n 2 gpurun_out/snap_dense.log | cut -c1-300
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
bash tools/prof_step.sh graph_r06 $B --steps 6 --warmup 3 > /dev/null 2>&1
python tools/prof_gaps.py gpurun_out/prof_graph_r06 --last-steps 4 | tee gpurun_out/gaps_graph_r06.txt
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r06a $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06a --last-steps 4 > gpurun_out/prof_serial_r06a/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r06a/summary4.txt > gpurun_out/step_breakdown_serial_r06a.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06a --last-steps 4 --by-grid > gpurun_out/step_by_grid_r06a.txt 2>&1
python tools/prof_gaps.py gpurun_out/prof_serial_r06a --last-steps 4 > gpurun_out/gaps_serial_r06a.txt
head -30 gpurun_out/step_breakdown_serial_r06a.txt
for v in base prio1 prio2 prio3 base prio1; do
  L=""; [ $v != base ] && L=tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/prio_variants.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
rm -f gpurun_out/snap384/snap_a.npz gpurun_out/snap384/product_a.npz
du -sh gpurun_out
