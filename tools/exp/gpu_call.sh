cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_conv_gpu.py tests/test_layers_gpu.py tests/test_loss_gpu.py -q 2>&1 | tail -1
timeout 200 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c90-200
