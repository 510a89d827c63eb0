set -x
cd $GRAFT_REPO_ROOT
timeout 120 tools/exp/mfma_split_exp > gpurun_out/mfma_split_exp.txt 2>&1
cat gpurun_out/mfma_split_exp.txt
timeout 600 bash tools/prof_loss.sh r02final --batches 256 --iters 10 > gpurun_out/prof_loss_r02final.log 2>&1
tail -5 gpurun_out/prof_loss_r02final.log
timeout 600 python bench.py > gpurun_out/bench_r02_n1.json 2> gpurun_out/bench_r02_n1.err
cat gpurun_out/bench_r02_n1.json
