#!/bin/bash
# round 6, call 13: bisect the configs[1] regression: the test alone (28 s), each of this round's changes on the torch-backend path reverted by a switch
set -u
cd $GRAFT_REPO_ROOT
T=tests/test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot
run() {
  rm -f /tmp/curve.txt
  env "$@" CD_AMD_PARITY_CURVES=/tmp/curve.txt timeout 600 python -m pytest $T -m gpu -q -x > /tmp/t.log 2>&1
  r=$(tail -1 /tmp/t.log | cut -c1-40)
  echo "$* | $r | $(sed -n '4,8p' /tmp/curve.txt | awk '{printf "e%s mean %s ckpt %s; ", $1, $3, $7}')"
}
for rep in 1 2; do
  run CD_DBG_NONE=1
  run CD_DBG_ZERO=aten
  run CD_DBG_SCALE=aten
  run CD_DBG_ROOT=plain
  run CD_DBG_MAT=1
  run CD_DBG_JOINT=old
  run CD_DBG_ZERO=aten CD_DBG_SCALE=aten CD_DBG_ROOT=plain CD_DBG_MAT=1 CD_DBG_JOINT=old
done 2>&1 | tee gpurun_out/config1_bisect.txt
