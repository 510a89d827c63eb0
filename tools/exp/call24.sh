#!/bin/bash
mkdir -p gpurun_out/r3
timeout 500 python bench.py > gpurun_out/r3/bench_final.json 2> gpurun_out/r3/bench_final.log
tail -2 gpurun_out/r3/bench_final.log
timeout 700 bash tools/prof_bench.sh r03f > gpurun_out/r3/prof_bench_final.log 2>&1
# keep the summaries, drop the raw rocprofv3 databases (gpurun copies back at most 64 MiB)
find gpurun_out/prof_r03f -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
du -sh gpurun_out; ls gpurun_out/prof_r03f
