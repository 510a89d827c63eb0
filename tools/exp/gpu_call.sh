set -x
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 900 python -m pytest tests/test_midas_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -30
cp gpurun_out/parity_log.txt gpurun_out/parity_midas.txt 2>/dev/null
