#!/bin/bash
# round 6, call 4: full GPU suite (early in the round), loss call after the fingerprint, host cost of a replayed step on both clips
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/gpu_suite_r06a.txt 2>&1
tail -n 16 gpurun_out/gpu_suite_r06a.txt
python tools/loss_bench.py --batches 256,1024 --iters 40 --warm 100 --brief 2>&1 | tail -2 | tee gpurun_out/loss_bench_r06c4.txt
python tools/host_cost.py --frames 244 2>/dev/null | tail -1 | tee gpurun_out/host_cost_244.json
python tools/host_cost.py --frames 1000 2>/dev/null | tail -1 | tee gpurun_out/host_cost_1000.json
