#!/bin/bash
# round 6, call 14: bisect the configs[1] regression over this round's commits: each commit's tree built on the box, its own test 4 times
set -u
cd $GRAFT_REPO_ROOT
T=tests/test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot
for c in 494b32d 45b5457 c1878f6 af41506; do
  ( cd tools/exp/bisect/$c && python -m consistent_depth_amd.build_native > /dev/null 2>&1
    for rep in 1 2 3 4; do
      rm -f /tmp/curve.txt
      CD_AMD_PARITY_CURVES=/tmp/curve.txt timeout 600 python -m pytest $T -m gpu -q -x > /tmp/t.log 2>&1
      echo "$c | $(tail -1 /tmp/t.log | cut -c1-30) | $(sed -n '4,7p;20,21p' /tmp/curve.txt | awk '{printf "e%s mean %s ckpt %s; ", $1, $3, $7}')"
    done )
done 2>&1 | tee gpurun_out/config1_bisect_commits.txt
