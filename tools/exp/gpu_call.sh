cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 600 python -m pytest tests/test_layers_gpu.py tests/test_conv_gpu.py -x -q -k "wgrad or weight_gradient" 2>&1 | tail -15
for a in fp32 split; do
for i in 0 1 2 3 4 5 7 9; do timeout 120 python tools/conv_bench.py --wgrad --arith $a --only $i 2>/dev/null; done
done > gpurun_out/wgrad_split_bench.txt 2>&1
cat gpurun_out/wgrad_split_bench.txt
