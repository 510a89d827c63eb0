"""Geometric-consistency loss on the fused gfx950 kernel.

Mirrors the operator surface of the reference's loss/consistency_loss.py
(/root/reference/loss/consistency_loss.py:92-253): `ConsistencyLoss(opt)(depths, metadata)`
-> (scalar loss attached to autograd, {"reprojection": (B,), "disparity": (B,)}).
Where the reference launches ~150 ATen kernels (pixel_grid, rays, baddbmm/bmm, project,
grid_sample, four masked means, and their autograd backward), this calls ONE fused
forward+backward pass through the C ABI (cd_consistency_loss_fwd_bwd,
include/consistent_depth_amd.h); the backward of the autograd node only scales the
already-computed gradient.
"""
from __future__ import annotations

import torch

from .. import _native

DEPTH_IDENTITY, DEPTH_EXP, DEPTH_RECIPROCAL = 0, 1, 2


def _prep(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def mask_sums(mask_fwd: torch.Tensor, mask_bwd: torch.Tensor) -> torch.Tensor:
    """S[b,k] = sum(mask_k[b]) -- dataset constants, cache them per pair (cd_mask_sums)."""
    mask_fwd, mask_bwd = _prep(mask_fwd), _prep(mask_bwd)
    B, _, H, W = mask_fwd.shape
    out = torch.empty(B, 2, dtype=torch.float32, device=mask_fwd.device)
    rc = _native.lib().cd_mask_sums(_native.dev_ptr(mask_fwd, "mask_fwd"), _native.dev_ptr(mask_bwd, "mask_bwd"),
                                    B, H, W, _native.dev_ptr(out), _native.stream_ptr(out.device))
    _native.check(rc, "cd_mask_sums")
    return out


def tile_windows(flows, masks) -> torch.Tensor:
    """Opaque per-tile source windows of the gradient kernel for B pairs -> uint8 (B, bytes_per_pair).
    Dataset constants like the mask sums: cache per pair, gather rows per batch (cd_tile_windows)."""
    f0, f1, m0, m1 = _prep(flows[0]), _prep(flows[1]), _prep(masks[0]), _prep(masks[1])
    B, _, H, W = f0.shape
    lib = _native.lib()
    nbytes = lib.cd_tile_windows_bytes(B, H, W)
    out = torch.empty(B, nbytes // B, dtype=torch.uint8, device=f0.device)
    rc = lib.cd_tile_windows(_native.dev_ptr(f0, "flows[0]"), _native.dev_ptr(f1, "flows[1]"),
                             _native.dev_ptr(m0, "masks[0]"), _native.dev_ptr(m1, "masks[1]"), B, H, W,
                             out.data_ptr(), _native.stream_ptr(out.device))
    _native.check(rc, "cd_tile_windows")
    return out


def _launch(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, msum, twin, intr, extr, lambda_r, lambda_b, mode, want_grad):
    lib = _native.lib()
    B, N, H, W = depth.shape
    if N != 2:
        raise ValueError(f"depths must be (B, 2, H, W) frame pairs, got N={N}")
    dev = depth.device
    for name, t, shape in (("flows[0]", flow_fwd, (B, 2, H, W)), ("flows[1]", flow_bwd, (B, 2, H, W)),
                           ("masks[0]", mask_fwd, (B, 1, H, W)), ("masks[1]", mask_bwd, (B, 1, H, W)),
                           ("intrinsics", intr, (B, 2, 4)), ("extrinsics", extr, (B, 2, 3, 4))):
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
    args = [_native.dev_ptr(depth, "depths"), _native.dev_ptr(flow_fwd, "flows[0]"),
            _native.dev_ptr(flow_bwd, "flows[1]"), _native.dev_ptr(mask_fwd, "masks[0]"),
            _native.dev_ptr(mask_bwd, "masks[1]"), _native.dev_ptr(msum, "mask_sums") if msum is not None else None]
    intr_p, extr_p = _native.dev_ptr(intr, "intrinsics"), _native.dev_ptr(extr, "extrinsics")     # (every operand is checked before anything is allocated)
    ws_bytes = lib.cd_consistency_loss_workspace_bytes(B, H, W)
    # (ABI 9: a loss workspace is initialised ONCE after allocation -- its header holds the row sweep's finished-workgroup counter)
    ws = _native.workspace("consistency_loss", ws_bytes, dev, on_new=lambda buf: _native.check(
        lib.cd_consistency_loss_workspace_init(buf.data_ptr(), buf.numel(), _native.stream_ptr(dev)), "cd_consistency_loss_workspace_init"))
    out = torch.empty(2 * B + 1, dtype=torch.float32, device=dev)
    reproj, disp, total = out[:B], out[B:2 * B], out[2 * B:]
    tail = [intr_p, extr_p, float(lambda_r), float(lambda_b), int(mode), B, H, W,
            reproj.data_ptr(), disp.data_ptr(), total.data_ptr()]
    if want_grad:
        grad = torch.empty_like(depth)
        tw = None
        if twin is not None:
            if twin.dtype != torch.uint8 or not twin.is_cuda or not twin.is_contiguous() or \
                    twin.numel() != lib.cd_tile_windows_bytes(B, H, W):
                raise ValueError("tile_windows: expected the contiguous uint8 result of tile_windows() for this batch")
            tw = twin.data_ptr()
        rc = lib.cd_consistency_loss_fwd_bwd(*args, tw, *tail, grad.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _native.stream_ptr(dev))
        _native.check(rc, "cd_consistency_loss_fwd_bwd")
    else:
        grad = None
        rc = lib.cd_consistency_loss_fwd(*args, *tail, ws.data_ptr(), ws.numel(), _native.stream_ptr(dev))
        _native.check(rc, "cd_consistency_loss_fwd")
    return total.reshape(()), reproj, disp, grad


class _FusedConsistency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, msum, twin, intr, extr, lambda_r, lambda_b, mode):
        total, reproj, disp, grad = _launch(depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd, msum, twin, intr, extr,
                                            lambda_r, lambda_b, mode, True)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(reproj, disp)
        ctx.set_materialize_grads(False)       # (autograd would fill two (B,) zero tensors per step for the non-differentiable outputs)
        return total, reproj, disp

    @staticmethod
    def backward(ctx, g_total, _g_reproj, _g_disp):
        (grad,) = ctx.saved_tensors
        if g_total is None:       # `total` did not take part in what was differentiated
            return (None,) * 12
        # d loss / d depth = (the kernel's gradient of `total`) x (upstream scalar, on the device): one hand-written launch (cd_eltwise op 3)
        out = torch.empty_like(grad)
        g = g_total.detach().to(torch.float32).reshape(1).contiguous()
        rc = _native.lib().cd_eltwise(grad.data_ptr(), g.data_ptr(), out.data_ptr(), grad.numel(), 3, _native.stream_ptr(grad.device))
        _native.check(rc, "cd_eltwise")
        return (out,) + (None,) * 11


def consistency_loss(depths, flows, masks, intrinsics, extrinsics, lambda_reprojection, lambda_view_baseline,
                     mask_sums=None, depth_mode=DEPTH_IDENTITY, tile_windows=None):
    """Functional form.  `depths` (B,2,H,W) is the depth itself, or -- with depth_mode
    DEPTH_EXP / DEPTH_RECIPROCAL -- the raw network output whose exp / reciprocal is the depth
    (mannequin_challenge_model.py:66 / midas_v2_model.py:67 fused into the kernel)."""
    depth = _prep(depths)
    f0, f1, m0, m1 = _prep(flows[0]), _prep(flows[1]), _prep(masks[0]), _prep(masks[1])
    intr, extr = _prep(intrinsics), _prep(extrinsics)
    msum = _prep(mask_sums) if mask_sums is not None else None
    if depth.requires_grad and torch.is_grad_enabled():
        return _FusedConsistency.apply(depth, f0, f1, m0, m1, msum, tile_windows, intr, extr,
                                       float(lambda_reprojection), float(lambda_view_baseline), int(depth_mode))
    total, reproj, disp, _ = _launch(depth, f0, f1, m0, m1, msum, None, intr, extr, lambda_reprojection,
                                     lambda_view_baseline, depth_mode, False)
    return total, reproj, disp


class ConsistencyLoss(torch.nn.Module):
    """Drop-in for the reference's ConsistencyLoss (same constructor, same call, same outputs).

    `opt` needs .lambda_reprojection and .lambda_view_baseline.  Extension (not in the
    reference): metadata["geometry_consistency"]["mask_sums"] (B,2) and ["tile_windows"], if present,
    are used as cached dataset constants (normaliser / source windows of the gradient kernel); `depth_mode` lets a model hand over its raw output.
    """

    def __init__(self, opt, depth_mode: int = DEPTH_IDENTITY):
        super().__init__()
        self.opt = opt
        self.depth_mode = depth_mode

    def __call__(self, depths, metadata):
        geom = metadata["geometry_consistency"]
        total, reproj, disp = consistency_loss(
            depths, geom["flows"], geom["masks"], metadata["intrinsics"], metadata["extrinsics"],
            self.opt.lambda_reprojection, self.opt.lambda_view_baseline,
            mask_sums=geom.get("mask_sums"), depth_mode=self.depth_mode, tile_windows=geom.get("tile_windows"))
        return total, {"reprojection": reproj, "disparity": disp}
