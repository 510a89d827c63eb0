mkdir -p gpurun_out; rm -f gpurun_out/parity_log.txt
timeout 900 python -m pytest tests/test_loss_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/loss_tests_r02e.txt
cat gpurun_out/loss_tests_r02e.txt
for v in "--variant 4 --pxt 2" "--variant 4 --pxt 4"; do
  timeout 300 python tools/loss_bench.py --batches 4,256,1024 --iters 10 $v 2>&1 | tail -3
done > gpurun_out/loss_bench_r02e.txt
cat gpurun_out/loss_bench_r02e.txt
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/prof_r02e; mkdir -p $OUT; R=$PWD; cd /tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python $R/tools/loss_bench.py --batches 256 --iters 5 --variant 4 --pxt 2 > $OUT/pmc_$name.log 2>&1
done
cd $R; python tools/prof_summary.py $OUT loss_sweep > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
