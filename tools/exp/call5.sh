mkdir -p gpurun_out/r3
( timeout 400 python -m pytest tests/test_loss_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r3/loss_tests2.txt 2>&1
for v in gp1 default; do
  if [ $v = default ]; then unset CD_AMD_LIB; else export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_$v.so; fi
  echo "== $v" >> gpurun_out/r3/loss_bench3.txt
  timeout 200 python tools/loss_bench.py --batches 256,1024 --iters 30 --variant 4 2>/dev/null >> gpurun_out/r3/loss_bench3.txt
done
unset CD_AMD_LIB
bash tools/prof_step.sh r3_midas --model midas2 --height 384 --width 384 --batch-size 8 --steps 3 --warmup 2 --no-loss-microbench --frames 20 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r3_midas --last-steps 2 > gpurun_out/r3/prof_midas_summary.txt 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r3_midas --last-steps 2 --by-grid > gpurun_out/r3/prof_midas_bygrid.txt 2>&1
rm -rf gpurun_out/prof_r3_midas/trace
cat gpurun_out/r3/loss_tests2.txt gpurun_out/r3/loss_bench3.txt; head -45 gpurun_out/r3/prof_midas_summary.txt; head -70 gpurun_out/r3/prof_midas_bygrid.txt
