cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 5 7 9 12 13 11 15; do timeout 120 python tools/conv_bench.py --only $i --cfgs 4x1,16x1,4x2,16x2 2>/dev/null | cut -c1-180; done
for i in 5 7; do timeout 120 python tools/conv_bench.py --dgrad --only $i --cfgs 4x1,16x1,4x2,16x2 2>/dev/null | cut -c1-180; done
timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c90-200
