"""Thin Python face of the MFMA convolution entry points (C ABI: cd_conv2d_*).
Tensors are NCHW fp32 on the HIP device; `(tensor, channel offset)` pairs address channel slices
of concat buffers in place."""
from __future__ import annotations

import torch

from .. import _native

KERNEL_SIZES = (1, 3, 5, 7, 11)


def pack_weights(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """w (Cout, Cin, k, k) -> packed filter (forward, or the flipped/transposed dgrad filter)."""
    Cout, Cin, k, k2 = w.shape
    assert k == k2 and k in KERNEL_SIZES, f"kernel size {k} not supported"
    lib = _native.lib()
    n = lib.cd_conv2d_packed_weight_floats(Cout, Cin, k, int(transposed))
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    wc = w.detach().contiguous()
    rc = lib.cd_conv2d_pack_weights(_native.dev_ptr(wc, "weight"), Cout, Cin, k, int(transposed), out.data_ptr(),
                                    _native.stream_ptr(w.device))
    _native.check(rc, "cd_conv2d_pack_weights")
    return out


def conv2d(x, packed_w, Cin, Cout, ks, bias=None, x_coff=0, out=None, y_coff=0, in_scale=None, in_shift=None,
           in_relu=False, stats=None, accumulate=False):
    """out[:, y_coff:y_coff+Cout] = conv(act(x[:, x_coff:x_coff+Cin])) + bias.  Returns `out`."""
    N, x_ctot, H, W = x.shape
    if out is None:
        out = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device)
    y_ctot = out.shape[1]
    opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else None  # noqa: E731
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.is_cuda and stats.is_contiguous() and stats.numel() == 2 * y_ctot
    rc = _native.lib().cd_conv2d_fwd(
        _native.dev_ptr(x, "x"), x_ctot, x_coff, Cin, _native.dev_ptr(packed_w, "packed_w"), opt(bias, "bias"),
        opt(in_scale, "in_scale"), opt(in_shift, "in_shift"), int(in_relu), _native.dev_ptr(out, "out"), y_ctot, y_coff,
        Cout, stats.data_ptr() if stats is not None else None, int(accumulate), N, H, W, ks, _native.stream_ptr(x.device))
    _native.check(rc, "cd_conv2d_fwd")
    return out


def wgrad_workspace_floats(Cout, Cin, ks) -> int:
    return _native.lib().cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks)


def wgrad_workspace(Cout, Cin, ks, device) -> torch.Tensor:
    return torch.empty(wgrad_workspace_floats(Cout, Cin, ks), dtype=torch.float32, device=device)


def conv2d_wgrad(x, dy, Cin, Cout, ks, dw, workspace, x_coff=0, dy_coff=0, in_scale=None, in_shift=None,
                 in_relu=False, accumulate=False, prezeroed=False):
    """dw (Cout,Cin,ks,ks) (+)= sum dy[:, dy_coff:+Cout] * act(x[:, x_coff:+Cin]) shifted by the taps."""
    N, x_ctot, H, W = x.shape
    opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else None  # noqa: E731
    rc = _native.lib().cd_conv2d_wgrad(
        _native.dev_ptr(x, "x"), x_ctot, x_coff, Cin, opt(in_scale, "in_scale"), opt(in_shift, "in_shift"), int(in_relu),
        _native.dev_ptr(dy, "dy"), dy.shape[1], dy_coff, Cout, _native.dev_ptr(dw, "dw"), int(accumulate) | (2 if prezeroed else 0),
        _native.dev_ptr(workspace, "workspace"), N, H, W, ks, _native.stream_ptr(x.device))
    _native.check(rc, "cd_conv2d_wgrad")
    return dw
