#!/bin/bash
# round 6, call 15: configs[1] with the gradient buffer zeroed by a library kernel (not hipMemsetAsync): the test alone, 8 times;
# then the tests that use cd_zero_bytes (engine arenas) and the bench
set -u
cd $GRAFT_REPO_ROOT
T=tests/test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot
for rep in 1 2 3 4 5 6 7 8; do
  rm -f /tmp/curve.txt
  CD_AMD_PARITY_CURVES=/tmp/curve.txt timeout 600 python -m pytest $T -m gpu -q -x > /tmp/t.log 2>&1
  echo "zero-kernel | $(tail -1 /tmp/t.log | cut -c1-30) | $(sed -n '4,6p;19,22p' /tmp/curve.txt | awk '{printf "e%s mean %s ckpt %s; ", $1, $3, $7}')"
done 2>&1 | tee gpurun_out/config1_zero_kernel.txt
( timeout 1200 python -m pytest tests/test_hourglass_engine_gpu.py tests/test_finetune_gpu.py tests/test_optim_gpu.py "tests/test_loop_gpu.py::test_full_length_run_vs_fp64_and_vs_the_reference_fp32_run" -m gpu -q -x 2>&1 | tail -3 )
grep -h "burn_in_state_bitwise" gpurun_out/parity_log.txt | tail -2 | cut -c1-120
python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
