#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c26
for i in 1 2; do
( time timeout 900 python -m pytest tests/test_loop_gpu.py -q -x -k "config1" ) > gpurun_out/c26/config1_$i.txt 2>&1; tail -6 gpurun_out/c26/config1_$i.txt | cut -c1-300
done
