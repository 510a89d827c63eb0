mkdir -p gpurun_out/r3
for v in i32gp2 default; do
  if [ $v = default ]; then unset CD_AMD_LIB; else export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_$v.so; fi
  echo "== $v" >> gpurun_out/r3/loss_bench4.txt
  timeout 200 python tools/loss_bench.py --batches 256,1024 --iters 30 --variant 4 2>&1 | grep -v amdgpu >> gpurun_out/r3/loss_bench4.txt
done
unset CD_AMD_LIB
bash tools/prof_loss.sh r03_i32 --batches 256 --iters 10 --variant 4 > /dev/null 2>&1
rm -rf gpurun_out/prof_r03_i32/trace gpurun_out/prof_r03_i32/pmc_*/
cat gpurun_out/r3/loss_bench4.txt; grep -A40 "loss_sweep_kernel" gpurun_out/prof_r03_i32/summary.txt | head -120
