cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --no-loss-microbench 2>/dev/null | tail -1 | cut -c1-330
CD_AMD_ENGINE_STREAMS=none timeout 600 bash tools/prof_step.sh r02split --backend hip --steps 4 --warmup 2 --no-cpu-baseline --no-loss-microbench --graph 0 > /dev/null 2>&1
python tools/prof_families.py gpurun_out/prof_r02split/summary.txt 2>&1 | head -60
