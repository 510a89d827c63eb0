set -x
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -25
cp gpurun_out/parity_log.txt gpurun_out/parity_conv_split.txt 2>/dev/null
for a in fp32 split; do
for i in 0 1 2 3 5 7; do timeout 120 python tools/conv_bench.py --arith $a --only $i --cfgs 16x1,8x1,4x1,16x2,8x2 ; done
for i in 0 1 3; do timeout 120 python tools/conv_bench.py --dgrad --arith $a --only $i --cfgs 16x1,8x1,16x2,8x2 ; done
done > gpurun_out/conv_split_bench.txt 2>&1
cat gpurun_out/conv_split_bench.txt
