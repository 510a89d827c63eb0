cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py -x -q 2>&1 | tail -5
grep conv_pointwise gpurun_out/parity_log.txt | tail -12
for a in split split1x1 split; do CD_AMD_CONV_ARITH=$a timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c90-200; done
