#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c27
export CD_AMD_REPORT=1
export AMD_LOG_LEVEL=1
( time timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/c27/gpu_suite.txt 2>&1
tail -n 12 gpurun_out/c27/gpu_suite.txt | cut -c1-300
dmesg 2>/dev/null | tail -20 > gpurun_out/c27/dmesg.txt
