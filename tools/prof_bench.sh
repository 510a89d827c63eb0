#!/bin/bash
# rocprofv3 evidence for the default bench.py command: kernel trace (+stats) and, in SEPARATE runs, PMC passes.
set -u
TAG=${1:-bench}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
export CD_AMD_CONV_TUNE_CACHE=$OUT/conv_tune.json
# first an unprofiled run that measures the convolution launch shapes into the cache, so the profiled runs below contain
# only the kernels of real steps; the profiled runs are eager (--graph 0): same kernels, individually traceable
python $REPO/bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 2 --warmup 3 > $OUT/tune.log 2>&1
CMD="python $REPO/bench.py --no-cpu-baseline --no-config5 --graph 0 --steps 10 --warmup 3"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-32)
  timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- $CMD --no-loss-microbench --steps 3 --warmup 2 > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/prof_step_summary.py $OUT --last-steps 10 > $OUT/summary_trace.txt 2>&1
python tools/prof_families.py $OUT/summary_trace.txt > $OUT/summary_families.txt 2>&1
# the loss kernels of the HBM-saturating micro-benchmark (bench.py's `roofline` object) run after the steps: list them unfiltered
(echo; echo '# fused loss kernels, all launches of the run, by grid (the B=256 rows are the roofline micro-benchmark):'; python tools/prof_step_summary.py $OUT --by-grid | grep -E 'loss_|tile_window') >> $OUT/summary_trace.txt 2>&1
python tools/prof_bench_pmc.py $OUT > $OUT/summary_pmc.txt 2>&1
cat $OUT/summary_families.txt; head -40 $OUT/summary_pmc.txt
