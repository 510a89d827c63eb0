mkdir -p gpurun_out/r3
( time timeout 900 python -m pytest tests/test_loop_gpu.py tests/test_dp_gpu.py tests/test_midas_gpu.py tests/test_masks_gpu.py tests/test_optim_gpu.py tests/test_warp_gpu.py tests/test_abi.py -x -q --durations=8 2>&1 | tail -30 ) > gpurun_out/r3/suite_tail2.txt 2>&1
bash tools/prof_bench.sh r03 > gpurun_out/r3/prof_bench_r03.txt 2>&1
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh r03_serial --graph 0 --steps 4 --warmup 3 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r03_serial --last-steps 4 > gpurun_out/r3/step_serial_r03_trace.txt 2>&1
python tools/prof_families.py gpurun_out/r3/step_serial_r03_trace.txt > gpurun_out/r3/step_serial_r03_families.txt 2>&1
timeout 500 python tools/conv_sweep.py --iters 5 > gpurun_out/r3/conv_sweep_r03.txt 2> gpurun_out/r3/conv_sweep_r03.err
timeout 400 python tools/wgrad_sweep.py --iters 5 > gpurun_out/r3/wgrad_sweep_r03.txt 2> gpurun_out/r3/wgrad_sweep_r03.err
rm -rf gpurun_out/prof_r03/trace gpurun_out/prof_r03/pmc_*/ gpurun_out/prof_r03_serial/trace
cat gpurun_out/r3/suite_tail2.txt; cat gpurun_out/r3/step_serial_r03_families.txt | head -30; tail -2 gpurun_out/r3/conv_sweep_r03.txt; tail -2 gpurun_out/r3/wgrad_sweep_r03.txt | cut -c1-300
