"""Test infrastructure: the fp64 CPU reference of the hourglass at the BASELINE batch (8 x 3 x 384 x 224) -- forward, every
parameter gradient, BatchNorm running statistics -- for tests/test_hourglass_engine_gpu.py::test_engine_matches_autograd[baseline_8x384x224].

It costs ~5 minutes of host time and tens of GB of host memory (torch's double convolution unfolds the whole batch), more than any
other test.  Round 3 computed it in a BACKGROUND process on the GPU box while the other tests ran; in round 4 three GPU boxes in a
row were lost during the CPU-reference phases of the suite, so the reference is now a COMMITTED GOLDEN
(tests/golden/engine_ref_8x384x224.npz), computed once in the build container from the same seeds the test uses:

    python tests/bg_reference.py --golden          (fp64; ~4 minutes with oracle/conv64.py)

stored in fp32 and SAMPLED (every tensor -- the prediction, each parameter gradient -- as flat[::stride], stride = max(1, numel //
2048): the test's measures are relative-L1 distances per tensor, which a regular sample of >= 2048 elements estimates to a few
per cent; the test samples the engine's tensors the same way, `sample()` below).  CD_AMD_TEST_LIVE_ENGINE_REF=1 brings the live
computation back (background process in a full session, inline when the test runs on its own):

    python tests/bg_reference.py <out.npz> [threads]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

N, H, W, SEED = 8, 384, 224, 0
GOLDEN = os.path.join(REPO, "tests", "golden", "engine_ref_8x384x224.npz")
SAMPLE = 2048


def stride_of(numel: int) -> int:
    return max(1, int(numel) // SAMPLE)


def sample(t):
    """flat[::stride]: numpy array or torch tensor -> 1-D of the same kind."""
    flat = t.reshape(-1)
    return flat[::stride_of(flat.shape[0])]


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


def write_golden():
    import torch
    from oracle import conv64
    orig = torch.nn.Conv2d._conv_forward

    def conv_forward(self, inp, weight, bias):      # the dgemm formulation of the fp64 convolution (5x faster on the build container)
        k = weight.shape[-1]
        if (inp.dtype == torch.float64 and self.stride == (1, 1) and self.dilation == (1, 1) and self.groups == 1 and
                weight.shape[-2] == k and k % 2 == 1 and tuple(self.padding) == ((k - 1) // 2,) * 2):
            return conv64._Conv64.apply(inp, weight, bias)
        return orig(self, inp, weight, bias)
    torch.nn.Conv2d._conv_forward = conv_forward
    try:
        res = compute()
    finally:
        torch.nn.Conv2d._conv_forward = orig
    out = {"sampled": np.array(SAMPLE)}
    for k, v in res.items():
        out[k] = (v if k.startswith("stat:") else sample(v)).astype(np.float32)
    np.savez_compressed(GOLDEN, **out)
    print("wrote", GOLDEN, os.path.getsize(GOLDEN) / 1e6, "MB,", len(out), "arrays")


def load_golden():
    """The committed reference as {key: array} (keys: pred, grad:<parameter>, stat:<buffer>, sampled), or None."""
    if not os.path.exists(GOLDEN):
        return None
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


if __name__ == "__main__":
    if sys.argv[1] == "--golden":
        write_golden()
        sys.exit(0)
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
