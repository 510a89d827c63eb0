#!/bin/bash
# serial (one stream, eager) kernel-time breakdown per family, batched wgrad on/off
set -u
export CD_AMD_CONV_TUNE_CACHE=$PWD/gpurun_out/conv_tune.json
python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 2 --warmup 3 > /dev/null 2>&1
for b in 1 0; do
  CD_AMD_WGRAD_BATCH=$b CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh wb$b --backend hip --steps 4 --warmup 3 --graph 0 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
  python tools/prof_step_summary.py gpurun_out/prof_wb$b --last-steps 4 > gpurun_out/prof_wb$b/summary4.txt 2>&1
  python tools/prof_families.py gpurun_out/prof_wb$b/summary4.txt > gpurun_out/fam_wb$b.txt 2>&1
  rm -rf gpurun_out/prof_wb$b/trace
  echo "== batch=$b"; head -14 gpurun_out/fam_wb$b.txt; tail -1 gpurun_out/fam_wb$b.txt
done
