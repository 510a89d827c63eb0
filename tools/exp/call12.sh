mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_layers_gpu.py tests/test_conv_gpu.py tests/test_midas_gpu.py tests/test_hourglass_engine_gpu.py -k "not baseline_8x384x224 and not fullres" -x -q 2>&1 | tail -6 ) > gpurun_out/r3/wg_tests.txt 2>&1
for m in 0 1; do
  echo "== CD_AMD_WGRAD_COT1=$m" >> gpurun_out/r3/wg_ab.txt
  CD_AMD_WGRAD_COT1=$m timeout 400 python tools/wgrad_sweep.py --iters 5 2>/dev/null | grep '"shape": \[[0-9]*, [0-9]*, 3,\|per_step' | cut -c1-110 >> gpurun_out/r3/wg_ab.txt
  CD_AMD_WGRAD_COT1=$m timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-config5 --no-loss-microbench 2>&1 >/dev/null | grep "timed region" >> gpurun_out/r3/wg_ab.txt
  CD_AMD_WGRAD_COT1=$m timeout 300 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 5 --warmup 2 --no-loss-microbench --frames 20 2>&1 >/dev/null | grep "timed region" >> gpurun_out/r3/wg_ab.txt
done
cat gpurun_out/r3/wg_tests.txt gpurun_out/r3/wg_ab.txt
