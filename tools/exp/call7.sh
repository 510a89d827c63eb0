mkdir -p gpurun_out/r3
( timeout 600 python -m pytest tests/test_midas_gpu.py tests/test_loss_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r3/midas_tests2.txt 2>&1
timeout 200 python tools/loss_bench.py --batches 256,1024 --iters 30 --variant 4 2>&1 | grep -v amdgpu > gpurun_out/r3/loss_bench5.txt
timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 5 --warmup 2 --no-loss-microbench --frames 20 > gpurun_out/r3/bench_midas_c.json 2> gpurun_out/r3/bench_midas_c.err
bash tools/prof_step.sh r3_midas2 --model midas2 --height 384 --width 384 --batch-size 8 --steps 3 --warmup 2 --no-loss-microbench --frames 20 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r3_midas2 --last-steps 2 > gpurun_out/r3/prof_midas2_summary.txt 2>&1
rm -rf gpurun_out/prof_r3_midas2/trace
cat gpurun_out/r3/midas_tests2.txt gpurun_out/r3/loss_bench5.txt; tail -2 gpurun_out/r3/bench_midas_c.err; head -30 gpurun_out/r3/prof_midas2_summary.txt
