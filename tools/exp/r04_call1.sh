#!/bin/bash
# round 4, GPU call 1: burn-in snapshot at 384x224 (-> fp64 golden on the CPU), baseline bench with in-run PMC traffic,
# stream-mode A/B under graph replay, the new GPU tests.
set -u
OUT=gpurun_out/r04_1; mkdir -p $OUT
export CD_AMD_CONV_TUNE_CACHE=$PWD/$OUT/conv_tune.json
( time timeout 600 python -m oracle.gen_golden_loop_384 snapshot gpurun_out/snap384 ) > $OUT/snapshot.log 2>&1
ls -la gpurun_out/snap384 >> $OUT/snapshot.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --no-config5 > $OUT/bench_default.json 2> $OUT/bench_default.log
for mode in level both none; do
  CD_AMD_ENGINE_STREAMS=$mode timeout 200 python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 30 > $OUT/bench_streams_$mode.json 2> $OUT/bench_streams_$mode.log
done
timeout 900 python -m pytest tests/test_driver_gpu.py tests/test_dp_gpu.py tests/test_midas_gpu.py -q -x -k "logged_at or parameter_only or bench_main or pooled_layer or golden_vectors" > $OUT/tests_new.log 2>&1
timeout 600 python -m pytest tests/test_loss_gpu.py -q -x > $OUT/tests_loss.log 2>&1
tail -3 $OUT/tests_new.log $OUT/tests_loss.log $OUT/snapshot.log; cat $OUT/bench_*.json | cut -c1-400
du -sh gpurun_out
