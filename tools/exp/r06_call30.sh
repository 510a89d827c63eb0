#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c30
export CD_AMD_REPORT=1
timeout 1500 python -m pytest tests/test_hourglass_engine_gpu.py tests/test_finetune_gpu.py tests/test_loop_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "not config1" > gpurun_out/c30/tests.txt 2>&1; tail -4 gpurun_out/c30/tests.txt | cut -c1-200
grep -n "burn_in_state_bitwise" gpurun_out/parity_log.txt | cut -c1-160 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 40 --warmup 10 --no-config5 --no-cpu-baseline --no-loss-microbench > gpurun_out/c30/bench_$i.json 2>gpurun_out/c30/bench_$i.err; cut -c90-200 gpurun_out/c30/bench_$i.json; done
