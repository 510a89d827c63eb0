mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_driver_gpu.py tests/test_dp_gpu.py tests/test_finetune_gpu.py -m gpu -q -x --durations=6 -k "not run_level and not short_finetune" 2>&1 | tail -25 > gpurun_out/tests_r02i.txt
cat gpurun_out/tests_r02i.txt
