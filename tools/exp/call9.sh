mkdir -p gpurun_out/r3
( time timeout 1500 python -m pytest tests/test_hourglass_engine_gpu.py tests/test_layers_gpu.py tests/test_loop_gpu.py tests/test_loss_gpu.py tests/test_masks_gpu.py tests/test_midas_gpu.py tests/test_optim_gpu.py tests/test_warp_gpu.py -k "not baseline_8x384x224" -x -q --durations=12 2>&1 | tail -40 ) > gpurun_out/r3/suite_tail.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err
CD_AMD_BN_MODE=normalize timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/r3/bench_c_norm.json 2> gpurun_out/r3/bench_c_norm.err
cat gpurun_out/r3/suite_tail.txt; tail -2 gpurun_out/r3/bench_c.err; tail -1 gpurun_out/r3/bench_c_norm.err
