mkdir -p gpurun_out/r3
( timeout 600 python -m pytest tests/test_midas_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r3/midas_tests.txt 2>&1
( timeout 400 python -m pytest tests/test_hourglass_engine_gpu.py -k inception_block -q 2>&1 | tail -12 ) > gpurun_out/r3/block_tests2.txt 2>&1
timeout 300 python tools/exp/diag_block.py > gpurun_out/r3/diag_block2.txt 2>&1
timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 5 --warmup 2 --no-loss-microbench --frames 20 > gpurun_out/r3/bench_midas_b.json 2> gpurun_out/r3/bench_midas_b.err
cat gpurun_out/r3/midas_tests.txt gpurun_out/r3/block_tests2.txt; grep -v amdgpu gpurun_out/r3/diag_block2.txt; tail -3 gpurun_out/r3/bench_midas_b.err; cut -c1-300 gpurun_out/r3/bench_midas_b.json
