"""Test infrastructure: the fp64 CPU reference of the hourglass at the BASELINE batch (8 x 3 x 384 x 224) takes ~5 minutes of host
time -- more than any other test.  conftest.py starts this script as a BACKGROUND process when the GPU session begins, so the
reference is computed on idle host cores while the other GPU tests run; test_engine_matches_autograd[baseline_8x384x224] then
loads the result (or computes it inline when it is run on its own).

    python tests/bg_reference.py <out.npz> [threads]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

N, H, W, SEED = 8, 384, 224, 0


def inputs():
    """(state dict fp64, x, dpred): what test_engine_matches_autograd builds for this shape, from the same seed."""
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    torch.manual_seed(SEED)
    ref = HourglassModel().double()
    HourglassModel()          # the test builds its fp32 twin here: keep the random stream aligned
    x = torch.rand(N, 3, H, W, dtype=torch.float64)
    dpred = torch.randn(N, 1, H, W, dtype=torch.float64)
    return ref, x, dpred


def compute():
    ref, x, dpred = inputs()
    ref.train()
    pred, _ = ref(x)
    pred.backward(dpred)
    out = {"pred": pred.detach().numpy()}
    for name, p in ref.named_parameters():
        if p.grad is not None:
            out["grad:" + name] = p.grad.numpy()
    for k, v in ref.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["stat:" + k] = v.numpy()
    return out


if __name__ == "__main__":
    dst = sys.argv[1]
    import psutil
    if psutil.virtual_memory().available < 48e9:     # fp64 autograd of 8 images keeps ~25 GB of activations, peaks higher (exit 3 = the test FAILS loudly)
        sys.exit(3)
    import torch
    if len(sys.argv) > 2:
        torch.set_num_threads(int(sys.argv[2]))
    res = compute()
    np.savez(dst + ".tmp.npz", **res)
    os.replace(dst + ".tmp.npz", dst)
