mkdir -p gpurun_out/r3
rm -f gpurun_out/parity_log.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 2>&1 | tail -60 ) > gpurun_out/r3/full_gpu_suite.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err
timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 5 --warmup 2 --no-loss-microbench --frames 20 > gpurun_out/r3/bench_midas_d.json 2> gpurun_out/r3/bench_midas_d.err
cat gpurun_out/r3/full_gpu_suite.txt; tail -3 gpurun_out/r3/bench_b.err; tail -2 gpurun_out/r3/bench_midas_d.err
