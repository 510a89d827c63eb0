cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_warp_gpu.py tests/test_masks_gpu.py tests/test_layers_gpu.py tests/test_optim_gpu.py -q 2>&1 | tail -2
