mkdir -p gpurun_out/r3
timeout 600 python tools/exp/diag_block.py > gpurun_out/r3/diag_block.txt 2>&1
for v in gp1 rec1 rec2; do
  export CD_AMD_LIB=$PWD/tools/exp/variants/libcd_amd_$v.so
  echo "== $v" >> gpurun_out/r3/loss_bench2.txt
  timeout 200 python tools/loss_bench.py --batches 256,1024 --iters 30 --variant 4 2>/dev/null >> gpurun_out/r3/loss_bench2.txt
done
unset CD_AMD_LIB
cat gpurun_out/r3/diag_block.txt gpurun_out/r3/loss_bench2.txt
