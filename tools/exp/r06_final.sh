#!/bin/bash
# round 6 evidence from ONE box: full GPU suite, default bench line, rocprofv3 of bench (kernel trace + PMC), serial family breakdown,
# the loss call's kernel trace + PMC, loss micro-bench, the streaming reference, gaps of a replayed step, host cost
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_20ep_r06.txt
rm -f gpurun_out/parity_log.txt $CD_AMD_PARITY_CURVES
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r06.txt 2>&1
tail -n 6 gpurun_out/gpu_suite_r06.txt
unset CD_AMD_PARITY_CURVES
python bench.py > gpurun_out/bench_r06_n1.json 2> gpurun_out/bench_r06_n1.log
tail -c 600 gpurun_out/bench_r06_n1.log
python tools/loss_bench.py --batches 4,32,256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -4 > gpurun_out/loss_bench_r06.txt
tools/hbm_stream 256 40 > gpurun_out/hbm_stream_256_final.txt; tools/hbm_stream 1024 20 > gpurun_out/hbm_stream_1024_final.txt
bash tools/prof_bench.sh r06 > gpurun_out/prof_bench_r06.log 2>&1
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r06 $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06 --last-steps 4 > gpurun_out/prof_serial_r06/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r06/summary4.txt > gpurun_out/step_breakdown_serial_r06.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r06 --last-steps 4 --by-grid > gpurun_out/step_kernels_by_grid_r06.txt 2>&1
python tools/prof_aten.py gpurun_out/prof_serial_r06 4 > gpurun_out/aten_in_step_final_r06.txt 2>&1
bash tools/prof_step.sh graph_r06f $B --steps 6 --warmup 3 > /dev/null 2>&1
python tools/prof_gaps.py gpurun_out/prof_graph_r06f --last-steps 4 > gpurun_out/gaps_graph_r06.txt 2>&1
bash tools/prof_loss.sh r06 --batches 256 --iters 40 --warm 100 > /dev/null 2>&1
python tools/host_cost.py --frames 244 2>/dev/null | tail -1 > gpurun_out/host_cost_r06.json
python tools/host_cost.py --frames 1000 2>/dev/null | tail -1 >> gpurun_out/host_cost_r06.json
rocm-smi --showclocks --showpower > gpurun_out/smi_r06.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out -type d -name "pmc_*" -prune -exec rm -rf {} + 2>/dev/null
head -30 gpurun_out/step_breakdown_serial_r06.txt; head -12 gpurun_out/prof_r06/summary.txt; cat gpurun_out/loss_bench_r06.txt
du -sh gpurun_out
# pack tables with 64 instead of 16 workgroups per descriptor (A/B, four alternations)
for rep in 1 2 3 4; do for v in base p64; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/variants/libcd_amd_$v.so
  CD_AMD_LIB=$L python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/p64_variants.txt
