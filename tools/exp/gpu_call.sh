mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hourglass_engine_gpu.py -m gpu -q -x -s -k "c_handle or eval_mode" 2>&1 | tail -30 > gpurun_out/tests_r02j.txt
cat gpurun_out/tests_r02j.txt
