#!/bin/bash
# round 6, call 11: launch shape 32 (8 row tiles + two channel chunks per barrier round): bit-identity tests, then A/B of the tuners with / without it
set -u
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_hourglass_engine_gpu.py -m gpu -q -x 2>&1 | tail -4 ) | tee gpurun_out/conv_tests_r06c11.txt
B="--backend hip --no-cpu-baseline --no-config5 --no-loss-microbench"
for rep in 1 2 3 4; do for v in 1 0; do
  CD_AMD_CONV_SHAPE32=$v python bench.py $B --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape32=$v', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/shape32_variants.txt
CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh s32 $B --steps 4 --warmup 3 --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_s32 --last-steps 4 --by-grid 2>&1 | grep "conv_fwd_split" | head -30 | cut -c1-200 | tee gpurun_out/s32_bygrid.txt
find gpurun_out -name "*.db" -delete; find gpurun_out -type d -name "trace" -prune -exec rm -rf {} + 2>/dev/null
