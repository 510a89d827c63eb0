#!/bin/bash
# rocprofv3 evidence for the fused loss kernel: kernel trace + PMC passes (each in its own run).
# usage (on the GPU box, from the repo root): bash tools/prof_loss.sh <tag> [loss_bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---batches 256 --iters 10}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $OLDPWD/tools/loss_bench.py $ARGS"
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $OLDPWD
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
