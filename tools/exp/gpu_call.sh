cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_loss_gpu.py -x -q 2>&1 | tail -3
timeout 120 python tools/loss_bench.py --batches 256 --iters 20 2>/dev/null | tail -4
timeout 120 python tools/loss_bench.py --batches 1024 --iters 10 2>/dev/null | tail -2
