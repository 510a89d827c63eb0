#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/gpu_suite_final.txt
tail -5 gpurun_out/r3/gpu_suite_final.txt
cp gpurun_out/parity_log.txt gpurun_out/r3/parity_log_final.txt 2>/dev/null
