"""TEST INFRASTRUCTURE ONLY -- a fast fp64 convolution for the CPU oracle.

torch's CPU `conv2d` has no vendor kernel for double: at 384x224 the fp64 hourglass runs at ~5 GFLOP/s on 8 cores (one
training step of the BASELINE batch = 5 minutes), which made fp64 ground truth at the headline shape a once-per-suite
luxury.  The same convolution as dense double-precision GEMMs (MKL / OpenBLAS dgemm, which IS fast): for every filter row
ky the padded input rows are unfolded along x into a (Cin*KW) x (H*W) matrix and multiplied by the (Cout) x (Cin*KW) slice of
the filter:

    y[n, co, :, :]  = b[co] + sum_ky  W[co, :, ky, :] (Cout x Cin*KW)  @  unfold_x(xpad[n, :, ky:ky+H, :]) (Cin*KW x H*W)
    dx              = the same convolution of dy with the filter flipped and transposed (stride 1, "same" padding)
    dW[:, :, ky, :] = sum_n  dy[n] (Cout x H*W)  @  unfold_x(xpad[n, :, ky:ky+H, :])^T

Pure restatement of torch.nn.functional.conv2d(x, w, b, padding=(k-1)//2) for stride 1 / odd square kernels -- checked against
it, values and gradients, in tests/test_oracle_properties_cpu.py.  Only used for float64 tensors (oracle/hourglass_ref.py
dispatches); fp32 keeps torch's mkldnn convolution.  Nothing here is reachable from consistent_depth_amd/.

OPT-IN (`ENABLED` / CD_ORACLE_CONV64=1, set by oracle/gen_golden_loop_384.py): measured 5.2x faster than torch's double conv2d on the
8-core build container (2 images of 384x224, forward + backward: 76 s -> 14.6 s), but 2x SLOWER in the one run on the GPU box
(tests/test_finetune_gpu.py's fp64 reference: ~250 s -> 544 s).  That run had OpenMP / MKL teams of 256 threads (the visible cores)
against a cgroup quota of 16 CPUs -- found afterwards, tests/conftest.py now caps the teams -- so the slowdown may have been dgemm
thrashing rather than the formulation; not re-measured with the capped teams, hence off by default on the GPU box.
"""
import os

import torch
import torch.nn.functional as F


def _cols(xp_n, ky, H, W, KW):
    """(Cin*KW, H*W) matrix of image n for filter row ky: rows (ci, kx), columns (y, x) -> xpad[ci, y + ky, x + kx]."""
    slab = xp_n[:, ky:ky + H, :]                       # (Cin, H, W + KW - 1)
    if KW == 1:
        return slab.reshape(slab.shape[0], H * W)
    return slab.unfold(2, KW, 1).permute(0, 3, 1, 2).reshape(slab.shape[0] * KW, H * W)   # one copy


def _conv_same(x, w, b):
    N, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    ph, pw = (KH - 1) // 2, (KW - 1) // 2
    xp = F.pad(x, (pw, pw, ph, ph)) if (ph or pw) else x
    y = x.new_empty(N, Cout, H, W)
    for n in range(N):
        acc = None
        for ky in range(KH):
            t = w[:, :, ky, :].reshape(Cout, Cin * KW) @ _cols(xp[n], ky, H, W, KW)
            acc = t if acc is None else acc.add_(t)
        if b is not None:
            acc += b.view(-1, 1)
        y[n] = acc.view(Cout, H, W)
    return y


class _Conv64(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return _conv_same(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_same(dy, w.flip(2, 3).transpose(0, 1).contiguous(), None)
        if ctx.needs_input_grad[1]:
            ph, pw = (KH - 1) // 2, (KW - 1) // 2
            xp = F.pad(x, (pw, pw, ph, ph)) if (ph or pw) else x
            dw = torch.zeros(Cout, Cin, KH, KW, dtype=w.dtype)
            for n in range(N):
                dyn = dy[n].reshape(Cout, H * W)
                for ky in range(KH):
                    dw[:, :, ky, :] += (dyn @ _cols(xp[n], ky, H, W, KW).t()).view(Cout, Cin, KW)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db


ENABLED = os.environ.get("CD_ORACLE_CONV64", "0") == "1"


def conv2d_same(x, w, b=None):
    """F.conv2d(x, w, b, padding=(k - 1) // 2), stride 1, odd square kernel; float64 on the CPU goes through dgemm."""
    k = w.shape[-1]
    if ENABLED and x.dtype == torch.float64 and x.device.type == "cpu" and w.shape[-2] == k and k % 2 == 1:
        return _Conv64.apply(x, w, b)
    return F.conv2d(x, w, b, padding=(k - 1) // 2)
