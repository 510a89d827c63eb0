cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_layers_gpu.py tests/test_conv_gpu.py -x -q -k "wgrad or weight_gradient" 2>&1 | tail -4
for a in split3 split; do for i in 16 17 18 6 10; do CD_AMD_CONV_ARITH=$a timeout 120 python tools/conv_bench.py --wgrad --only $i 2>/dev/null; done; done
timeout 120 python -m pytest tests/test_finetune_gpu.py tests/test_hourglass_engine_gpu.py -x -q -k "reproducible or c_handle or both_conv or 2x64x96" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c90-200
