cd $GRAFT_REPO_ROOT
export CD_AMD_REPORT=1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14
cp gpurun_out/parity_log.txt gpurun_out/parity_full_r02.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_r02_n1.json 2> gpurun_out/bench_r02_n1.err
cat gpurun_out/bench_r02_n1.json | cut -c1-330
