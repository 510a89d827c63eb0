"""Camera-geometry operators other stages of the reference reuse,
on the HIP kernels.  Mirrors the call signatures of /root/reference/utils/geometry.py:
`pixel_grid` :9-19 and `sample` :201-208 (the rest of that file is folded into the fused
loss kernel and has no separate entry point on the hot path)."""
from __future__ import annotations

import torch

from .. import _native


def pixel_grid(batch_size, shape, device=None):
    """(batch_size, 2, H, W) pixel positions (x, y), top-left (0, 0) -- expanded view, no copy."""
    H, W = shape
    device = device or torch.device("cuda", torch.cuda.current_device())
    x = torch.arange(W, dtype=torch.float32, device=device)
    y = torch.arange(H, dtype=torch.float32, device=device)
    grid = torch.stack((x[None, :].expand(H, W), y[:, None].expand(H, W)), 0)
    return grid[None].expand(batch_size, -1, -1, -1)


def sample(data, uv):
    """Bilinear border-padded sampling of data (B,C,H,W) at pixel coords uv (B,2,H,W)."""
    data = data.float().contiguous()
    uv = uv.float().contiguous()
    B, C, H, W = data.shape
    if tuple(uv.shape) != (B, 2, H, W):
        raise ValueError(f"uv: expected {(B, 2, H, W)}, got {tuple(uv.shape)}")
    out = torch.empty_like(data)
    rc = _native.lib().cd_sample_bilinear_border(_native.dev_ptr(data, "data"), _native.dev_ptr(uv, "uv"),
                                                 B, C, H, W, out.data_ptr(), _native.stream_ptr(data.device))
    _native.check(rc, "cd_sample_bilinear_border")
    return out
