#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c28
export CD_AMD_REPORT=1
( time timeout 1500 python -m pytest tests/test_loop_gpu.py -q -x -k "config1" -s ) > gpurun_out/c28/config1.txt 2>&1; tail -12 gpurun_out/c28/config1.txt | cut -c1-400
grep -n "config1" gpurun_out/parity_log.txt | tail -2 | cut -c1-600
