cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_layers_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -5
for a in fp32 split; do
for i in 11 14 15; do timeout 120 python tools/conv_bench.py --arith $a --only $i 2>/dev/null; timeout 120 python tools/conv_bench.py --wgrad --arith $a --only $i 2>/dev/null; done
timeout 120 python tools/conv_bench.py --dgrad --arith $a --only 8 2>/dev/null
done > gpurun_out/k3_bench.txt 2>&1
cat gpurun_out/k3_bench.txt
