mkdir -p gpurun_out/r3
rm -f gpurun_out/parity_log.txt
( time timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=15 2>&1 | tail -32 ) > gpurun_out/r3/full_gpu_suite2.txt 2>&1
timeout 500 python bench.py > gpurun_out/r3/bench_final.json 2> gpurun_out/r3/bench_final.err
bash tools/prof_bench.sh r03 > gpurun_out/r3/prof_bench_r03.txt 2>&1
rm -rf gpurun_out/prof_r03/trace gpurun_out/prof_r03/pmc_*/
cat gpurun_out/r3/full_gpu_suite2.txt; tail -4 gpurun_out/r3/bench_final.err; cut -c1-1500 gpurun_out/r3/bench_final.json
