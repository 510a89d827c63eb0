cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py -x -q 2>&1 | tail -3
for i in 0 1 2 3 5 7; do timeout 120 python tools/conv_bench.py --only $i --cfgs 16x1,8x1,4x1,8x4 2>/dev/null | cut -c1-120; done
for i in 0 1 3; do timeout 120 python tools/conv_bench.py --dgrad --only $i --cfgs 8x4,4x1 2>/dev/null | cut -c1-120; done
for i in 0 1 2 3 4 5 7 9; do timeout 120 python tools/conv_bench.py --wgrad --only $i 2>/dev/null; done
timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c90-200
