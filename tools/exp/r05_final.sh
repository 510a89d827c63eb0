#!/bin/bash
# end of round 5: the GPU suite, the default bench line, the loss call's rocprofv3 evidence, the streaming reference
set -u
cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
rm -f gpurun_out/parity_log.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 ) > gpurun_out/gpu_suite_r05b.txt 2>&1
tail -n 6 gpurun_out/gpu_suite_r05b.txt
python bench.py > gpurun_out/bench_r05b_n1.json 2> gpurun_out/bench_r05b_n1.log
tail -c 1500 gpurun_out/bench_r05b_n1.json
python tools/hbm_ref.py > gpurun_out/hbm_ref_r05.json 2>/dev/null
bash tools/prof_loss.sh r05b --batches 256 --iters 40 --warm 100 > /dev/null 2>&1
head -40 gpurun_out/prof_r05b/summary.txt
du -sh gpurun_out/prof_r05b
find gpurun_out/prof_r05b -name "*.db" -size +6M -delete
