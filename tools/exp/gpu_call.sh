mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py -m gpu -q -x --durations=5 2>&1 | tail -15 > gpurun_out/tests_r02h.txt
cat gpurun_out/tests_r02h.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-loss-microbench > gpurun_out/bench_r02h.json 2> gpurun_out/bench_r02h.err
tail -2 gpurun_out/bench_r02h.err
bash tools/prof_step.sh r02h > gpurun_out/prof_step_r02h.log 2>&1; tail -45 gpurun_out/prof_r02h/summary.txt
