"""Python face of the memory-bound layer kernels (cd_bn_*, cd_avgpool2_*, cd_upsample2x_*, ...).
All tensors NCHW fp32 on the HIP device; (tensor, channel offset, C) address channel slices."""
from __future__ import annotations

import torch

from .. import _native

_p = _native.dev_ptr


def _o(t, name="t"):
    return _native.dev_ptr(t, name) if t is not None else None


def _s(t):
    return _native.stream_ptr(t.device)


STAT_SLOTS = _native.BN_STAT_SLOTS


def new_stats(ctot, device) -> torch.Tensor:
    """Zeroed batch-statistics buffer (STAT_SLOTS, ctot, 2) fp64 for the conv epilogue / BatchNorm entry points."""
    return torch.zeros(STAT_SLOTS, ctot, 2, dtype=torch.float64, device=device)


def bn_normalize(x, coff, C, stats, mean_invstd, eps=1e-5, running_mean=None, running_var=None, momentum=0.1):
    N, ctot, H, W = x.shape
    assert stats.shape == (STAT_SLOTS, ctot, 2) and stats.is_contiguous(), "stats must be (STAT_SLOTS, ctot, 2)"
    rc = _native.lib().cd_bn_normalize(_p(x), ctot, coff, C, stats.data_ptr(), float(eps), _o(running_mean), _o(running_var),
                                       float(momentum), _p(mean_invstd), N, H, W, _s(x))
    _native.check(rc, "cd_bn_normalize")


def bn_finalize(stats, coff, C, count, mean_invstd, scale, shift, eps=1e-5, gamma=None, beta=None, running_mean=None,
                running_var=None, momentum=0.1):
    """stats (STAT_SLOTS,ctot,2) fp64 -> scale/shift (ctot,) for relu(raw*scale+shift) on load; no pass over the activation."""
    assert stats.shape[0] == STAT_SLOTS and stats.is_contiguous()
    rc = _native.lib().cd_bn_finalize(stats.data_ptr(), stats.shape[1], coff, C, float(count), float(eps), _o(gamma), _o(beta),
                                      _o(running_mean), _o(running_var), float(momentum), _p(mean_invstd), _p(scale), _p(shift),
                                      _native.stream_ptr(scale.device))
    _native.check(rc, "cd_bn_finalize")


def bn_relu_bwd(dA, d_coff, xhat, x_coff, C, mean_invstd, sums, gamma=None, beta=None, dgamma=None, dbeta=None,
                sums_prezeroed=False, scale=None, shift=None, overwrite_affine=False):
    """dgamma / dbeta are ACCUMULATED (the engine's convention for every parameter gradient) unless `overwrite_affine`."""
    N, d_ctot, H, W = dA.shape
    rc = _native.lib().cd_bn_relu_bwd(_p(dA), d_ctot, d_coff, _p(xhat), xhat.shape[1], x_coff, C, _o(gamma), _o(beta),
                                      _p(mean_invstd), _o(scale), _o(shift), sums.data_ptr(), int(bool(sums_prezeroed)) | (2 if overwrite_affine else 0), _o(dgamma),
                                      _o(dbeta), N, H, W,
                                      _s(dA))
    _native.check(rc, "cd_bn_relu_bwd")


def avgpool2_fwd(x, x_coff, C, y, y_coff=0, in_scale=None, in_shift=None, in_relu=False):
    N, x_ctot, H, W = x.shape
    rc = _native.lib().cd_avgpool2_fwd(_p(x), x_ctot, x_coff, _o(in_scale), _o(in_shift), int(in_relu), _p(y), y.shape[1],
                                       y_coff, C, N, H, W, _s(x))
    _native.check(rc, "cd_avgpool2_fwd")


def avgpool2_bwd(dy, dy_coff, dx, dx_coff, C, accumulate):
    N, dx_ctot, H, W = dx.shape
    rc = _native.lib().cd_avgpool2_bwd(_p(dy), dy.shape[1], dy_coff, _p(dx), dx_ctot, dx_coff, C, N, H, W, int(accumulate),
                                       _s(dx))
    _native.check(rc, "cd_avgpool2_bwd")


def upsample2x_add_fwd(lo, lo_coff, C, out, out_coff=0, hi=None, hi_coff=0, lo_relu=False, hi_relu=False, lo_scale=None,
                       lo_shift=None, hi_scale=None, hi_shift=None):
    N, lo_ctot, h, w = lo.shape
    rc = _native.lib().cd_upsample2x_add_fwd(_p(lo), lo_ctot, lo_coff, _o(lo_scale), _o(lo_shift), int(lo_relu), _o(hi),
                                             hi.shape[1] if hi is not None else 0, hi_coff, _o(hi_scale), _o(hi_shift),
                                             int(hi_relu), _p(out), out.shape[1], out_coff, C, N, h, w, _s(lo))
    _native.check(rc, "cd_upsample2x_add_fwd")


def upsample2x_bwd(dout, d_coff, dlo, l_coff, C, accumulate):
    N, l_ctot, h, w = dlo.shape
    rc = _native.lib().cd_upsample2x_bwd(_p(dout), dout.shape[1], d_coff, _p(dlo), l_ctot, l_coff, C, N, h, w,
                                         int(accumulate), _s(dlo))
    _native.check(rc, "cd_upsample2x_bwd")


def upsample2x_halfpixel_fwd(lo, lo_coff, C, out, out_coff=0):
    N, lo_ctot, h, w = lo.shape
    rc = _native.lib().cd_upsample2x_halfpixel_fwd(_p(lo), lo_ctot, lo_coff, _p(out), out.shape[1], out_coff, C, N, h, w, _s(lo))
    _native.check(rc, "cd_upsample2x_halfpixel_fwd")


def upsample2x_halfpixel_bwd(dout, d_coff, dlo, l_coff, C, accumulate):
    N, l_ctot, h, w = dlo.shape
    rc = _native.lib().cd_upsample2x_halfpixel_bwd(_p(dout), dout.shape[1], d_coff, _p(dlo), l_ctot, l_coff, C, N, h, w,
                                                   int(accumulate), _s(dlo))
    _native.check(rc, "cd_upsample2x_halfpixel_bwd")


class _BilinearUp2(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=...) on the hand-written kernels: forward one gather per output
    pixel, backward the adjoint as a gather per INPUT pixel (no atomics: ATen's backward scatters with atomics, ~1.3 ms per call at the
    MiDaS decoder's sizes).  align_corners=True: cd_upsample2x_add_fwd / cd_upsample2x_bwd (the hourglass' up-sampling);
    False: cd_upsample2x_halfpixel_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, align_corners):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise RuntimeError("bilinear_up2: fp32 (N, C, H, W) tensors on the HIP device (no CPU path)")
        x = x.contiguous()
        N, C, h, w = x.shape
        out = torch.empty((N, C, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        if align_corners:
            upsample2x_add_fwd(x, 0, C, out)
        else:
            upsample2x_halfpixel_fwd(x, 0, C, out)
        ctx.align_corners, ctx.in_shape = bool(align_corners), (N, C, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dx = torch.empty(ctx.in_shape, dtype=dout.dtype, device=dout.device)
        if ctx.align_corners:
            upsample2x_bwd(dout, 0, dx, 0, ctx.in_shape[1], False)
        else:
            upsample2x_halfpixel_bwd(dout, 0, dx, 0, ctx.in_shape[1], False)
        return dx, None


def bilinear_up2(x, align_corners):
    return _BilinearUp2.apply(x, align_corners)


def add_slice(src, s_coff, dst, d_coff, C, accumulate):
    N, d_ctot, H, W = dst.shape
    rc = _native.lib().cd_add_slice(_p(src), src.shape[1], s_coff, _p(dst), d_ctot, d_coff, C, N, H, W, int(accumulate),
                                    _s(dst))
    _native.check(rc, "cd_add_slice")


def set_layers_mode(bits):
    """Test / measurement hook (cd_debug_set_layers_mode): bit 0 = the scalar streaming kernels of rounds 1-5."""
    _native.check(_native.lib().cd_debug_set_layers_mode(int(bits)), "cd_debug_set_layers_mode")


def channel_sum(src, coff, C, out, accumulate=False):
    N, ctot, H, W = src.shape
    rc = _native.lib().cd_channel_sum(_p(src), ctot, coff, C, N, H, W, _p(out), int(accumulate), _s(src))
    _native.check(rc, "cd_channel_sum")
