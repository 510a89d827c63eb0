mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_driver_gpu.py tests/test_dp_gpu.py tests/test_masks_gpu.py tests/test_finetune_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/tests_r02f.txt
cat gpurun_out/tests_r02f.txt
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err
tail -5 gpurun_out/bench_r02f.err; cat gpurun_out/bench_r02f.json
