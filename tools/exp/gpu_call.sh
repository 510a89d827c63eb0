cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_hourglass_engine_gpu.py tests/test_finetune_gpu.py tests/test_midas_gpu.py -x -q -k "not baseline_8x384x224" 2>&1 | tail -15
cp gpurun_out/parity_log.txt gpurun_out/parity_split_engine.txt 2>/dev/null
for a in fp32 split; do CD_AMD_CONV_ARITH=$a timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1; done > gpurun_out/bench_arith.txt
cat gpurun_out/bench_arith.txt
