#!/bin/bash
set -u
export CD_AMD_CONV_TUNE_CACHE=$PWD/gpurun_out/conv_tune.json
timeout 900 python -m pytest tests/test_driver_gpu.py tests/test_loop_gpu.py -q -x 2>&1 | tail -4
timeout 600 python tools/exp/val_sweep_time.py 2>&1 | grep "validation sweep"
