mkdir -p gpurun_out/r3
( timeout 400 python -m pytest tests/test_loss_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r3/loss_tests.txt 2>&1
for v in old default nostatic gp1; do
  case $v in
    old) export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_old.so; unset CD_AMD_SWEEP_STATIC_GEO;;
    default) unset CD_AMD_LIB; unset CD_AMD_SWEEP_STATIC_GEO;;
    nostatic) unset CD_AMD_LIB; export CD_AMD_SWEEP_STATIC_GEO=0;;
    gp1) export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_gp1.so; unset CD_AMD_SWEEP_STATIC_GEO;;
  esac
  echo "== $v" >> gpurun_out/r3/loss_bench.txt
  timeout 200 python tools/loss_bench.py --batches 256,1024 --iters 20 --variant 4 >> gpurun_out/r3/loss_bench.txt 2>&1
done
unset CD_AMD_LIB; unset CD_AMD_SWEEP_STATIC_GEO
( timeout 300 python -m pytest tests/test_hourglass_engine_gpu.py -k inception_block -x -q 2>&1 | tail -12 ) > gpurun_out/r3/block_tests.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err
timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 3 --warmup 2 --no-loss-microbench --frames 20 > gpurun_out/r3/bench_midas_a.json 2> gpurun_out/r3/bench_midas_a.err
cat gpurun_out/r3/loss_tests.txt gpurun_out/r3/loss_bench.txt gpurun_out/r3/block_tests.txt; tail -3 gpurun_out/r3/bench_a.err; cat gpurun_out/r3/bench_a.json | cut -c1-600; tail -5 gpurun_out/r3/bench_midas_a.err; cat gpurun_out/r3/bench_midas_a.json | cut -c1-400
