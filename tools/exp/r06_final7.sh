#!/bin/bash
# round 6: the full GPU suite on the final tree (configs[1] continuations in a child process) + smoke
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
export CD_AMD_PARITY_CURVES=$PWD/gpurun_out/parity_20ep_r06d.txt
rm -f gpurun_out/parity_log.txt $CD_AMD_PARITY_CURVES
( time timeout 2700 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r06d.txt 2>&1
tail -n 14 gpurun_out/gpu_suite_r06d.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
