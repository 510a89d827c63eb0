"""Child process of tests/test_loop_gpu.py::test_config1_torch_convs_hip_loss_from_the_same_snapshot: the continuations of the burn-in
snapshot (RUNS of them, one after the other) with the hourglass convolutions on PyTorch-ROCm / MIOpen (BASELINE configs[1]) and the HIP loss / Adam.

    python tests/config1_worker.py <job.pt>

job.pt (torch.save): snap (state / m1 / m2 / k, CPU tensors), clip (make_synthetic_dataset arguments), paths (one per run), K, T, n_pairs, plans
({epoch: [[pair, ...], ...]}), pair_order.  Writes every run's artefacts under its path's output directory and prints one OUT_DIR= line per finished run.

Why a child: the third-party half of this configuration (MIOpen's immediate-mode kernels under torch's autograd, replayed ~600 times)
ended ONE of the ~10 full-suite runs of round 6 with SIGABRT inside torch.cuda.synchronize() -- no message from the runtime, not
reproducible alone (2 of 2 clean) or in the next full run -- and an abort inside the pytest process takes the whole suite's results with
it.  In a child the parent sees the signal, reports it and repeats the runs that did not finish, once."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))      # make_synthetic_dataset


def main():
    import torch
    job = torch.load(sys.argv[1], weights_only=False)
    os.environ["CD_AMD_MC_BACKEND"] = "torch"
    torch.backends.cudnn.benchmark = False
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.params import Video3dParamsParser
    snap, K, T, n_pairs = job["snap"], job["K"], job["T"], job["n_pairs"]
    pair_id = {tuple(p): i for i, p in enumerate(job["pair_order"])}
    plans = job["plans"]
    for path in job["paths"]:           # the continuations one after the other (MIOpen compiles its kernels once per process)
        range_dir, _ = msd.write_dataset(path, **job["clip"])
        params = Video3dParamsParser().parse(["--path", path, "--num_epochs", str(K + T), "--batch_size", "4", "--print_freq", "0"])
        ft = DepthFineTuner(range_dir, list(range(job["clip"]["n_frames"])), params)
        assert ft.model.backend == "torch" and ft.model._engine is None
        ft.resume_from(snap["state"], snap["m1"], snap["m2"], snap["k"], epoch=K, total_iters=K * n_pairs)
        ft.epoch_plan = lambda e: [[pair_id[tuple(p)] for p in batch] for batch in plans[str(e)]]
        ft.fine_tune()
        assert [list(map(int, pr)) for pr in ft.store.pair_indices()] == job["pair_order"]
        torch.cuda.synchronize()
        print("OUT_DIR=" + ft.out_dir, flush=True)
        del ft
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
