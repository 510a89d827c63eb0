#!/bin/bash
set -u
export CD_AMD_REPORT=1
rm -f gpurun_out/parity_log.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/gpu_suite_r04.txt 2>&1
tail -n 30 gpurun_out/gpu_suite_r04.txt
