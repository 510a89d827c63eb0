cd $GRAFT_REPO_ROOT
timeout 600 python tools/conv_sweep.py --no-pipe-axis --iters 4 > gpurun_out/conv_sweep_r02.txt 2> gpurun_out/conv_sweep_r02.err
tail -3 gpurun_out/conv_sweep_r02.txt | cut -c1-300
timeout 300 python tools/wgrad_sweep.py --iters 4 > gpurun_out/wgrad_sweep_r02.txt 2> gpurun_out/wgrad_sweep_r02.err
tail -2 gpurun_out/wgrad_sweep_r02.txt
