#!/bin/bash
# round 4 evidence: step trace + PMC (bench.py --graph 0), one-stream family breakdown, loss kernel trace + PMC
set -u
export TMPDIR=/tmp
bash tools/prof_bench.sh r04 > gpurun_out/prof_bench_r04.log 2>&1
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh serial_r04 --backend hip --steps 4 --warmup 3 --graph 0 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r04 --last-steps 4 > gpurun_out/prof_serial_r04/summary4.txt 2>&1
python tools/prof_families.py gpurun_out/prof_serial_r04/summary4.txt > gpurun_out/step_breakdown_serial_r04.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_serial_r04 --last-steps 4 --by-grid > gpurun_out/step_by_grid_r04.txt 2>&1
bash tools/prof_loss.sh r04 --batches 256 --iters 10 > gpurun_out/prof_loss_r04.log 2>&1
python tools/loss_bench.py --batches 4,32,256,1024 --iters 20 > gpurun_out/loss_bench_r04.txt 2>&1
find gpurun_out -name "*.db" -delete
find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
find gpurun_out -type d -name "pmc_*" -prune -exec rm -rf {} + 2>/dev/null
du -sh gpurun_out; ls gpurun_out gpurun_out/prof_r04 gpurun_out/prof_loss_r04
