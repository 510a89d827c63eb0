#!/bin/bash
# round 6, call 12: is the configs[1] divergence the box / MIOpen or this round's code?  The same test, round 5's tree and this round's, on ONE box.
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
T=tests/test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot
( cd tools/exp/r05_tree && python -m consistent_depth_amd.build_native > /dev/null 2>&1; ls -la consistent_depth_amd/libcd_amd.so;
  CD_AMD_PARITY_CURVES=$GRAFT_REPO_ROOT/gpurun_out/config1_r05tree.txt timeout 900 python -m pytest $T -m gpu -q -x 2>&1 | tail -3 )
CD_AMD_PARITY_CURVES=$PWD/gpurun_out/config1_r06tree.txt timeout 900 python -m pytest $T -m gpu -q -x 2>&1 | tail -3
( cd tools/exp/r05_tree && CD_AMD_PARITY_CURVES=$GRAFT_REPO_ROOT/gpurun_out/config1_r05tree_b.txt timeout 900 python -m pytest $T -m gpu -q -x 2>&1 | tail -3 )
for f in gpurun_out/config1_r05tree.txt gpurun_out/config1_r06tree.txt gpurun_out/config1_r05tree_b.txt; do echo "== $f"; cut -c1-75 $f | sed -n 3,8p; done
