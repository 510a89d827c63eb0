#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
rm -f gpurun_out/parity_log.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r05c.txt 2>&1
tail -n 22 gpurun_out/gpu_suite_r05c.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
