mkdir -p gpurun_out/r3
( timeout 600 python -m pytest tests/test_hourglass_engine_gpu.py tests/test_conv_gpu.py tests/test_layers_gpu.py -k "not baseline_8x384x224 and not fullres" -x -q 2>&1 | tail -5 ) > gpurun_out/r3/engine_tests3.txt 2>&1
( CD_AMD_BN_APPLY=0 timeout 600 python -m pytest tests/test_hourglass_engine_gpu.py -k "2x64x96 or 2x32x48 or inception_block or eval_mode" -x -q 2>&1 | tail -5 ) >> gpurun_out/r3/engine_tests3.txt 2>&1
for m in 1 0 1 0; do
  CD_AMD_BN_APPLY=$m timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-config5 --no-loss-microbench 2>&1 >/dev/null | grep "timed region" | sed "s/^/BN_APPLY=$m /" >> gpurun_out/r3/bn_ab.txt
done
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh r03_serial2 --graph 0 --steps 4 --warmup 3 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r03_serial2 --last-steps 4 > gpurun_out/r3/step_serial_r03b_trace.txt 2>&1
python tools/prof_families.py gpurun_out/r3/step_serial_r03b_trace.txt > gpurun_out/r3/step_serial_r03b_families.txt 2>&1
CD_AMD_BN_APPLY=0 CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh r03_serial3 --graph 0 --steps 4 --warmup 3 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r03_serial3 --last-steps 4 > gpurun_out/r3/step_serial_r03c_trace.txt 2>&1
python tools/prof_families.py gpurun_out/r3/step_serial_r03c_trace.txt > gpurun_out/r3/step_serial_r03c_families.txt 2>&1
rm -rf gpurun_out/prof_r03_serial2/trace gpurun_out/prof_r03_serial3/trace
cat gpurun_out/r3/engine_tests3.txt gpurun_out/r3/bn_ab.txt; head -14 gpurun_out/r3/step_serial_r03b_families.txt; head -14 gpurun_out/r3/step_serial_r03c_families.txt
