cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_midas_gpu.py -x -q -k "not network and not finetune" 2>&1 | tail -4
for a in fp32 split; do
for i in 16 17 18 6 10; do timeout 120 python tools/conv_bench.py --arith $a --only $i --cfgs 4x1,4x2 2>/dev/null; done
for i in 9 10; do timeout 120 python tools/conv_bench.py --dgrad --arith $a --only $i --cfgs 4x1,4x2 2>/dev/null; done
done > gpurun_out/k1_bench.txt 2>&1
cat gpurun_out/k1_bench.txt
