"""Autograd faces of the hand-written block kernels (csrc/bn_block.hip) an autograd-driven backbone is built from -- the MiDaS v2 /
ResNeXt-101 network of BASELINE configs[4] (monodepth/midas_net.py):

    bn_act(x, bn, relu, res=None)   act(BatchNorm2d_train(x) [+ res])      cd_bn_block_fwd / cd_bn_block_bwd
    relu(x), add(a, b)              the decoder's element-wise pieces       cd_eltwise
    maxpool3s2(x)                   nn.MaxPool2d(3, 2, 1)                    cd_maxpool3s2_fwd / _bwd

fp32 NCHW tensors on the HIP device; no CPU path (the fp64 CPU twins of the tests use the ATen modules).  `bn` is the nn.BatchNorm2d that
owns the parameters and running statistics: same state_dict keys, same running-statistics update (momentum, unbiased variance,
num_batches_tracked) as the ATen module in training mode; in eval mode (running statistics) the ATen op is used -- fine-tuning runs the
network in training mode (reference: monodepth/midas_v2_model.py::train)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _native

_p = _native.dev_ptr


def _o(t):
    return _native.dev_ptr(t) if t is not None else None


def _chk(x, name):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise RuntimeError(f"{name}: fp32 (N, C, H, W) tensors on the HIP device (no CPU path)")
    x = x.contiguous()
    if x.data_ptr() % 16:      # a contiguous view at an odd storage offset: the kernels move 16 bytes per lane (cd_bn_block_* / cd_eltwise refuse it)
        x = x.clone()
    return x


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, bn, relu):
        x = _chk(x, "bn_act")
        N, C, H, W = x.shape
        res = _chk(res, "bn_act") if res is not None else None
        dev = x.device
        y = torch.empty_like(x)
        mi = torch.empty(C, 2, dtype=torch.float32, device=dev)
        scratch = torch.empty(2 * C, dtype=torch.float32, device=dev)
        stats = torch.empty(_native.BN_STAT_SLOTS * C * 2, dtype=torch.float64, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        if bn.momentum is not None:
            momentum = bn.momentum
        elif track and bn.num_batches_tracked is not None:
            # nn.BatchNorm2d(momentum=None): the CUMULATIVE moving average, factor 1 / num_batches_tracked after the increment
            # (torch/nn/modules/batchnorm.py).  One host read per forward: only this unusual configuration pays it.
            momentum = 1.0 / float(bn.num_batches_tracked.item())
        else:
            momentum = 0.0
        rc = _native.lib().cd_bn_block_fwd(_p(x), _o(gamma), _o(beta), _o(res), int(relu), _o(bn.running_mean) if track else None,
                                           _o(bn.running_var) if track else None, float(momentum), float(bn.eps), _p(y), _p(mi),
                                           _p(scratch[:C]), _p(scratch[C:]), stats.data_ptr(), C, N, H, W, _native.stream_ptr(dev))
        _native.check(rc, "cd_bn_block_fwd")
        ctx.relu, ctx.has_res, ctx.affine = bool(relu), res is not None, gamma is not None
        ctx.save_for_backward(x, y if relu else None, gamma, mi)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mi = ctx.saved_tensors
        dy = _chk(dy, "bn_act backward")
        N, C, H, W = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev) if ctx.affine else None
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        rc = _native.lib().cd_bn_block_bwd(_p(dy), _p(x), _o(y), _o(gamma), _p(mi), int(ctx.relu), _p(dx), _o(dres),
                                           _p(dgb[0]) if ctx.affine else None, _p(dgb[1]) if ctx.affine else None, sums.data_ptr(),
                                           C, N, H, W, _native.stream_ptr(dev))
        _native.check(rc, "cd_bn_block_bwd")
        return dx, (dgb[0] if ctx.affine else None), (dgb[1] if ctx.affine else None), dres, None, None


def bn_act(x, bn, relu, res=None):
    """act(bn(x) [+ res]) with act = ReLU (relu=True) or identity.  Training mode: the hand-written block (HIP device only: a CPU tensor raises); eval mode: ATen on the device."""
    _need_device(x, "bn_act")
    if bn.training and x.dtype == torch.float32:
        return _BnAct.apply(x, bn.weight, bn.bias, res, bn, relu)
    # eval mode (running statistics: inference before / after the fine-tuning, never the step) and non-fp32 tensors: the framework's ops
    y = bn(x)
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


class _Eltwise(torch.autograd.Function):
    """op 0: relu(a); op 1: a + b."""

    @staticmethod
    def forward(ctx, a, b, op):
        a = _chk(a, "eltwise")
        b = _chk(b, "eltwise") if b is not None else None
        y = torch.empty_like(a)
        rc = _native.lib().cd_eltwise(_p(a), _o(b), _p(y), a.numel(), int(op), _native.stream_ptr(a.device))
        _native.check(rc, "cd_eltwise")
        ctx.op = int(op)
        if op == 0:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.op == 1:
            return dy, dy, None
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        rc = _native.lib().cd_eltwise(_p(dy), _p(y), _p(dx), dy.numel(), 2, _native.stream_ptr(dy.device))
        _native.check(rc, "cd_eltwise (relu backward)")
        return dx, None, None


def _need_device(x, what):
    """The blocks of the hip back end have no CPU path (the twin for CPU work is MidasNet(backend="torch"))."""
    if not x.is_cuda:
        raise RuntimeError(f"ops.blocks.{what}: the hand-written block needs a tensor on the HIP device, got {x.device}")


def relu(x):
    _need_device(x, "relu")
    return _Eltwise.apply(x, None, 0) if (x.dtype == torch.float32 and x.dim() == 4) else F.relu(x)


def add(a, b):
    _need_device(a, "add")
    ok = a.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape
    return _Eltwise.apply(a, b, 1) if ok else a + b


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "maxpool3s2")
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(N, C, Ho, Wo, dtype=x.dtype, device=x.device)
        arg = torch.empty(N, C, Ho, Wo, dtype=torch.uint8, device=x.device)
        rc = _native.lib().cd_maxpool3s2_fwd(_p(x), _p(y), arg.data_ptr(), C, N, H, W, _native.stream_ptr(x.device))
        _native.check(rc, "cd_maxpool3s2_fwd")
        ctx.save_for_backward(arg)
        ctx.in_shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        N, C, H, W = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        rc = _native.lib().cd_maxpool3s2_bwd(_p(dy), arg.data_ptr(), _p(dx), C, N, H, W, _native.stream_ptr(dy.device))
        _native.check(rc, "cd_maxpool3s2_bwd")
        return dx


def maxpool3s2(x):
    _need_device(x, "maxpool3s2")
    return _MaxPool3s2.apply(x) if (x.dtype == torch.float32 and x.dim() == 4) else F.max_pool2d(x, 3, 2, 1)
