"""`HipConv2d`: an nn.Conv2d whose forward, input gradient and weight gradient run on the hand-written gfx950 MFMA kernels
(cd_conv2d_fwd / cd_conv2d_wgrad) -- the layer the MiDaS v2 backbone (BASELINE configs[4]) is built from.

Same parameters and state_dict keys as nn.Conv2d (a checkpoint loads unchanged).  Supported: square kernels 1/3/5/7/11,
"same" padding (k-1)/2, stride 1 or 2, any `groups`:
  * groups (ResNeXt's 32 x 8d 3x3): every group is a dense convolution on a channel slice of the SAME input / output
    buffers (the kernels address (tensor, channel offset, channels)), one launch per group -- correct and on the matrix
    cores; a fused grouped kernel is the known next step (at 8 channels per group a 16-wide MFMA tile is half empty);
  * stride 2: the stride-1 "same" output sampled at even positions (identical values; 4x the MACs of a strided kernel --
    only the stem, 3 bottlenecks and 3 down-sample 1x1 of ResNeXt-101 are strided).
Filters are re-packed by ONE table launch per forward (weights move under the optimiser), the packed buffers are
allocated once per layer.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _native
from . import conv as C


class _HipConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        Cout, cin_g, ks, _ = weight.shape
        G, s = layer.groups, layer.stride[0]
        cout_g = Cout // G
        lib, stream = _native.lib(), _native.stream_ptr(x.device)
        pk, _ = layer._packed(weight)
        y = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device)
        bptr = _native.dev_ptr(bias, "bias") if bias is not None else None
        for g in range(G):
            rc = lib.cd_conv2d_fwd(_native.dev_ptr(x, "x"), Cin, g * cin_g, cin_g, pk[g].data_ptr(),
                                   (bptr + 4 * g * cout_g) if bptr is not None else None, None, None, 0, y.data_ptr(), Cout, g * cout_g,
                                   cout_g, None, 0, N, H, W, ks, stream)
            _native.check(rc, "cd_conv2d_fwd")
        ctx.layer, ctx.hw = layer, (H, W)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y[:, :, ::s, ::s].contiguous() if s > 1 else y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        layer, (H, W) = ctx.layer, ctx.hw
        N, Cin = x.shape[:2]
        Cout, cin_g, ks, _ = weight.shape
        G, s = layer.groups, layer.stride[0]
        cout_g = Cout // G
        lib, stream = _native.lib(), _native.stream_ptr(x.device)
        if s > 1:   # adjoint of the sub-sampling: zeros between the samples
            dyf = torch.zeros(N, Cout, H, W, dtype=torch.float32, device=dy.device)
            dyf[:, :, ::s, ::s] = dy
        else:
            dyf = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _, pkT = layer._packed(weight, transposed_too=True)
            dx = torch.empty_like(x)
            for g in range(G):
                rc = lib.cd_conv2d_fwd(dyf.data_ptr(), Cout, g * cout_g, cout_g, pkT[g].data_ptr(), None, None, None, 0, dx.data_ptr(), Cin,
                                       g * cin_g, cin_g, None, 0, N, H, W, ks, stream)
                _native.check(rc, "cd_conv2d_fwd (dgrad)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws = layer._wgrad_workspace(cout_g, cin_g, ks, x.device)
            for g in range(G):
                rc = lib.cd_conv2d_wgrad(_native.dev_ptr(x, "x"), Cin, g * cin_g, cin_g, None, None, 0, dyf.data_ptr(), Cout, g * cout_g, cout_g,
                                         dw.data_ptr() + 4 * g * cout_g * cin_g * ks * ks, 0, ws.data_ptr(), N, H, W, ks, stream)
                _native.check(rc, "cd_conv2d_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None


class HipConv2d(torch.nn.Conv2d):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        k, s, p = self.kernel_size, self.stride, self.padding
        if not (k[0] == k[1] and k[0] in C.KERNEL_SIZES and s[0] == s[1] and s[0] in (1, 2) and p[0] == p[1] == (k[0] - 1) // 2
                and self.dilation == (1, 1) and self.padding_mode == "zeros"):
            raise ValueError(f"HipConv2d: unsupported geometry kernel {k} stride {s} padding {p}")
        self._pk = self._pkT = self._table = self._tableT = self._wptr = self._ws = None

    def _build(self, weight, transposed):
        """Packed buffers of every group (zeroed once: padding elements are never written) + the pack table."""
        lib = _native.lib()
        Cout, cin_g, ks, _ = weight.shape
        cout_g = Cout // self.groups
        oc, ic = (cin_g, cout_g) if transposed else (cout_g, cin_g)
        n = (lib.cd_conv2d_packed_weight_floats(oc, ic, ks, 0) + 63) // 64 * 64
        arena = torch.zeros(self.groups * n, dtype=torch.float32, device=weight.device)
        views = [arena[g * n:(g + 1) * n] for g in range(self.groups)]
        dt = np.dtype([("w", "<u8"), ("packed", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("tr", "<i4"),
                       ("OC", "<i4"), ("IC", "<i4"), ("oc_off", "<i4"), ("ic_off", "<i4")])
        tab = np.zeros(self.groups, dt)
        for g in range(self.groups):
            tab[g] = (weight.data_ptr() + 4 * g * cout_g * cin_g * ks * ks, views[g].data_ptr(), cout_g, cin_g, ks, int(transposed), oc, ic, 0, 0)
        return views, torch.from_numpy(tab.view(np.uint8).copy()).to(weight.device), arena

    def _packed(self, weight, transposed_too=False):
        if not (weight.is_cuda and weight.is_contiguous() and weight.dtype == torch.float32):
            raise RuntimeError("HipConv2d: weights must be contiguous fp32 on the HIP device (no CPU path)")
        if self._wptr != weight.data_ptr():       # first use, or the optimiser re-homed the parameter
            self._pk, self._table, self._arena = self._build(weight, False)
            self._pkT = self._tableT = None
            self._wptr = weight.data_ptr()
        lib, stream = _native.lib(), _native.stream_ptr(weight.device)
        if transposed_too:
            if self._pkT is None:
                self._pkT, self._tableT, self._arenaT = self._build(weight, True)
            _native.check(lib.cd_conv2d_pack_weights_table(self._tableT.data_ptr(), self.groups, stream), "cd_conv2d_pack_weights_table")
        else:
            _native.check(lib.cd_conv2d_pack_weights_table(self._table.data_ptr(), self.groups, stream), "cd_conv2d_pack_weights_table")
        return self._pk, self._pkT

    def _wgrad_workspace(self, cout_g, cin_g, ks, device):
        if self._ws is None:
            self._ws = C.wgrad_workspace(cout_g, cin_g, ks, device)
        return self._ws

    def forward(self, x):
        return _HipConvFn.apply(x, self.weight, self.bias, self)
