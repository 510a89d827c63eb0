cd $GRAFT_REPO_ROOT
CD_AMD_ENGINE_STREAMS=none timeout 300 bash tools/prof_step.sh r02final_serial --backend hip --steps 4 --warmup 2 --no-cpu-baseline --no-loss-microbench --graph 0 > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_r02final_serial --last-steps 4 > gpurun_out/step_sum_final.txt 2>&1
find gpurun_out/prof_r02final_serial -name "*.db" -delete
python tools/prof_families.py gpurun_out/step_sum_final.txt 2>&1 | head -24
