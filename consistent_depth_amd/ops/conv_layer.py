"""`HipConv2d`: an nn.Conv2d whose forward, input gradient and weight gradient run on the hand-written gfx950 MFMA kernels
(cd_conv2d_fwd / cd_conv2d_wgrad) -- the layer the MiDaS v2 backbone (BASELINE configs[4]) is built from.

Same parameters and state_dict keys as nn.Conv2d (a checkpoint loads unchanged).  Supported: square kernels 1/3/5/7/11,
"same" padding (k-1)/2, stride 1 or 2, any `groups`:
  * groups (ResNeXt's 32 x 8d 3x3): every group is a dense convolution on a channel slice of the SAME input / output
    buffers (the kernels address (tensor, channel offset, channels)); ALL groups run in ONE launch per pass
    (cd_conv2d_fwd_grouped / cd_conv2d_wgrad_grouped: the group is a grid dimension of the split-bf16 kernels) -- round 2
    issued one launch per group (32 x 33 bottlenecks x {forward, input gradient, weight gradient + unpack} = 4224 launches
    per step, which made the MiDaS step host-bound: 222 ms of enqueue for a 243 ms step).  At 8 channels per group a
    16-wide MFMA column tile is half empty;
  * stride 2: the stride-1 "same" output sampled at even positions (identical values; 4x the MACs of a strided kernel --
    only the stem, 3 bottlenecks and 3 down-sample 1x1 of ResNeXt-101 are strided).
  * 1x1, dense (the bottleneck entry / exit convolutions: 2/3 of ResNeXt-101's multiply-adds, at 12x12 .. 96x96 images with
    256 .. 2048 channels): a plain GEMM  Y[n] = W [Cout x Cin] . X[n] [Cin x HW]  -- not a stencil.  On the hand-written kernels like
    everything else (no library GEMM on the path; rounds 3-5 carried a torch.matmul / bmm route behind CD_AMD_MIDAS_1X1=gemm, removed in
    round 6): >= 512 channels on the chunked split-bf16 kernel csrc/conv1x1_split.hip::conv1x1_split_kc_kernel (~100 TFLOP/s
    fp32-equivalent where the staged fp32-MFMA kernel reached 40-77) and wgrad1x1_split, 96x96 planes with <= 384 input channels on
    the LDS-resident 1x1 kernel.
Filters are re-packed once per forward (weights move under the optimiser): a `PackPool` shared by the layers of a network
packs EVERY filter of the network, forward and transposed layouts, in ONE table launch (round 2: two launches of 16 workgroups
per layer and pass -- 53 ms of a 211 ms MiDaS step); a layer outside a pool packs its own filters.
"""
from __future__ import annotations

import numpy as np
import torch


from .. import _native
from . import conv as C


class PackPool:
    """All HipConv2d layers of one network: their filters (forward + transposed layouts, every group) are packed by ONE
    cd_conv2d_pack_weights_table launch per `run()` -- call it once per forward, before the first layer."""

    def __init__(self):
        self.layers, self._table, self._key, self.fresh = [], None, None, False

    def register(self, layer):
        self.layers.append(layer)
        layer._pool = self

    def run(self):
        layers = [l for l in self.layers if l._uses_packed()]
        key = tuple(l.weight.data_ptr() for l in layers)
        if key != self._key:      # first use, or the optimiser re-homed the parameters: rebuild the descriptors
            tabs = []
            for l in layers:
                w = l.weight
                if not (w.is_cuda and w.is_contiguous() and w.dtype == torch.float32):
                    raise RuntimeError("HipConv2d: weights must be contiguous fp32 on the HIP device (no CPU path)")
                l._pk, tab, l._arena = l._build(w, False)
                l._pkT, tabT, l._arenaT = l._build(w, True)
                l._wptr = w.data_ptr()
                tabs += [tab, tabT]
            self._table = torch.cat(tabs).contiguous()
            self._n = sum(2 * l.groups for l in layers)
            self._key = key
        if self._n > 65535:
            raise RuntimeError("PackPool: too many filters for one table launch")
        lib = _native.lib()
        _native.check(lib.cd_conv2d_pack_weights_table(self._table.data_ptr(), self._n, _native.stream_ptr(self._table.device)),
                      "cd_conv2d_pack_weights_table")
        self.fresh = True

    def invalidate(self):
        """The weights moved (an optimiser step): the packed copies are stale until the next run()."""
        self.fresh = False


def _dense_cfg(groups, ks, Cin, Cout, N, H, W, device):
    """(tile_rows, co_tiles) of the fastest launch shape of a DENSE k x k convolution (ops.conv.tuned_config: timed once per shape on
    scratch tensors, cached; the result does not depend on it), or None: grouped / 1x1 / tuning disabled (CD_AMD_MIDAS_CONV_TUNE=0) ->
    the library's rule.  Round 6: configs[4]'s decoder 3x3 convolutions had run on the rule's shape only."""
    import os
    if groups != 1 or ks < 3 or Cin < 8 or os.environ.get("CD_AMD_MIDAS_CONV_TUNE", "1") == "0":
        return None
    return C.tuned_config(ks, Cin, Cout, N, H, W, device)


_WGRAD_SIDE = {}


def _wgrad_side_stream(device):
    """The side stream of the weight gradients (one per device), or None under CD_AMD_MIDAS_WGRAD_STREAM=0."""
    import os
    if os.environ.get("CD_AMD_MIDAS_WGRAD_STREAM", "1") == "0":
        return None
    key = torch.device(device).index
    if key not in _WGRAD_SIDE:
        _WGRAD_SIDE[key] = torch.cuda.Stream(device=device)
    return _WGRAD_SIDE[key]


class _HipConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        Cout, cin_g, ks, _ = weight.shape
        G, s = layer.groups, layer.stride[0]
        ctx.full_hw = None
        if ks == 1 and s > 1:      # a strided 1x1 reads only the sampled pixels: sub-sample FIRST (exact, s^2 fewer multiply-adds)
            ctx.full_hw = tuple(x.shape[2:])
            x, s = x[:, :, ::s, ::s], 1
        x = x.contiguous()
        N, Cin, H, W = x.shape
        cout_g = Cout // G
        lib, stream = _native.lib(), _native.stream_ptr(x.device)
        pk, _ = layer._packed(weight)
        y = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device)
        bptr = _native.dev_ptr(bias, "bias") if bias is not None else None
        cfg = _dense_cfg(G, ks, Cin, Cout, N, H, W, x.device)
        if cfg is not None:     # dense k x k (the decoder): the launch shape timed once per shape, like the hourglass engine does
            C.conv2d(x, pk[0], Cin, Cout, ks, bias=bias, out=y, cfg=cfg)
        else:
            rc = lib.cd_conv2d_fwd_grouped(_native.dev_ptr(x, "x"), Cin, 0, cin_g, pk[0].data_ptr(), layer._pack_stride, bptr, y.data_ptr(), Cout, 0,
                                           cout_g, G, 0, N, H, W, ks, stream)
            _native.check(rc, "cd_conv2d_fwd_grouped")
        ctx.layer, ctx.hw, ctx.s = layer, (H, W), s
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y[:, :, ::s, ::s].contiguous() if s > 1 else y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        layer, (H, W) = ctx.layer, ctx.hw
        N, Cin = x.shape[:2]
        Cout, cin_g, ks, _ = weight.shape
        G, s = layer.groups, ctx.s
        cout_g = Cout // G
        lib, stream = _native.lib(), _native.stream_ptr(x.device)
        if s > 1:   # adjoint of the sub-sampling: zeros between the samples
            dyf = torch.zeros(N, Cout, H, W, dtype=torch.float32, device=dy.device)
            dyf[:, :, ::s, ::s] = dy
        else:
            dyf = dy.contiguous()
        dx = dw = db = None
        # The weight gradient and the input gradient of a layer are independent: the weight gradient is enqueued on a side stream forked
        # here and joined before this function returns, so that its workgroups fill the compute units the input gradient's last round of
        # workgroups leaves idle (288 workgroups of the 24x24 planes on 256 CUs).  Round 6, two alternations on one box: 72.9 / 73.1 ->
        # 73.6 / 73.7 pairs/s on BASELINE configs[4] (CD_AMD_MIDAS_WGRAD_STREAM=0: one stream).  Same results either way.
        side = _wgrad_side_stream(x.device) if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1]) else None
        if side is not None:
            dw = torch.empty_like(weight)
            ws, ws_stride = layer._wgrad_workspace(cout_g, cin_g, ks, x.device)
            cur = torch.cuda.current_stream(x.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            side.wait_event(fork)
            with torch.cuda.stream(side):
                rc = lib.cd_conv2d_wgrad_grouped(_native.dev_ptr(x, "x"), Cin, 0, cin_g, dyf.data_ptr(), Cout, 0, cout_g, G, dw.data_ptr(), 0,
                                                 ws.data_ptr(), ws_stride, N, H, W, ks, _native.stream_ptr(x.device))
                _native.check(rc, "cd_conv2d_wgrad_grouped")
        if ctx.needs_input_grad[0]:
            _, pkT = layer._packed(weight, transposed_too=True)
            dx = torch.empty_like(x)
            cfg = _dense_cfg(G, ks, Cout, Cin, N, H, W, x.device)
            if cfg is not None:
                C.conv2d(dyf, pkT[0], Cout, Cin, ks, out=dx, cfg=cfg)
            else:
                rc = lib.cd_conv2d_fwd_grouped(dyf.data_ptr(), Cout, 0, cout_g, pkT[0].data_ptr(), layer._pack_strideT, None, dx.data_ptr(), Cin, 0,
                                               cin_g, G, 0, N, H, W, ks, stream)
                _native.check(rc, "cd_conv2d_fwd_grouped (dgrad)")
            if ctx.full_hw is not None:      # (strided 1x1: the input was sub-sampled first)
                st = layer.stride[0]
                full = torch.zeros((N, Cin) + ctx.full_hw, dtype=dx.dtype, device=dx.device)
                full[:, :, ::st, ::st] = dx
                dx = full
        if side is not None:
            done = torch.cuda.Event()
            done.record(side)
            torch.cuda.current_stream(x.device).wait_event(done)
        elif ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            ws, ws_stride = layer._wgrad_workspace(cout_g, cin_g, ks, x.device)
            rc = lib.cd_conv2d_wgrad_grouped(_native.dev_ptr(x, "x"), Cin, 0, cin_g, dyf.data_ptr(), Cout, 0, cout_g, G, dw.data_ptr(), 0,
                                             ws.data_ptr(), ws_stride, N, H, W, ks, stream)
            _native.check(rc, "cd_conv2d_wgrad_grouped")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from .layers import channel_sum      # (hand-written reduction: cd_channel_sum)
            dyc = dy.contiguous()
            db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
            channel_sum(dyc, 0, Cout, db)
        return dx, dw, db, None


class HipConv2d(torch.nn.Conv2d):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        k, s, p = self.kernel_size, self.stride, self.padding
        if not (k[0] == k[1] and k[0] in C.KERNEL_SIZES and s[0] == s[1] and s[0] in (1, 2) and p[0] == p[1] == (k[0] - 1) // 2
                and self.dilation == (1, 1) and self.padding_mode == "zeros"):
            raise ValueError(f"HipConv2d: unsupported geometry kernel {k} stride {s} padding {p}")
        self._pk = self._pkT = self._table = self._tableT = self._wptr = self._ws = self._pool = None

    def _uses_packed(self):
        return True

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
        if transposed:
            self._pack_strideT = n
        else:
            self._pack_stride = n
        return views, torch.from_numpy(tab.view(np.uint8).copy()).to(weight.device), arena

    def _packed(self, weight, transposed_too=False):
        if not (weight.is_cuda and weight.is_contiguous() and weight.dtype == torch.float32):
            raise RuntimeError("HipConv2d: weights must be contiguous fp32 on the HIP device (no CPU path)")
        if self._pool is not None:
            # packed by the pool's single launch at the start of this forward.  The pool is fresh from that launch until the
            # optimiser moves the weights (engine.FineTuneStep -> model.weights_updated() -> PackPool.invalidate()); a pooled layer
            # called outside the network's forward after an update (a sub-module, a feature extractor, a test) re-packs the pool.
            if not (self._pool.fresh and self._wptr == weight.data_ptr()):
                self._pool.run()
            return self._pk, self._pkT
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
        """(buffer, floats per group): every group's per-workgroup partial sums side by side."""
        if self._ws is None:
            n = (_native.lib().cd_conv2d_wgrad_workspace_floats(cout_g, cin_g, ks) + 63) // 64 * 64
            self._ws = (torch.empty(self.groups * n, dtype=torch.float32, device=device), n)
        return self._ws

    def forward(self, x):
        return _HipConvFn.apply(x, self.weight, self.bias, self)
