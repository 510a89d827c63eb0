#!/bin/bash
for i in 1 2 3; do timeout 300 python tools/exp/wgrad_dbg.py 2>&1 | grep "bad entries" | tr '\n' ' '; echo; done
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "weight_gradient" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hourglass_engine_gpu.py -x -q -m gpu -k "inception_block" 2>&1 | tail -2
