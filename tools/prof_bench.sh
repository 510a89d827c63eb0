#!/bin/bash
# rocprofv3 evidence for the default bench.py command: kernel trace (+stats) and, in SEPARATE runs, PMC passes.
set -u
TAG=${1:-bench}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 3"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-32)
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- $CMD --no-loss-microbench --steps 3 --warmup 2 > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/prof_step_summary.py $OUT > $OUT/summary_trace.txt 2>&1
python tools/prof_bench_pmc.py $OUT > $OUT/summary_pmc.txt 2>&1
head -30 $OUT/summary_trace.txt; head -40 $OUT/summary_pmc.txt
