"""Camera-geometry operators other stages of the reference reuse,
on the HIP kernels.  Mirrors the call signatures of /root/reference/utils/geometry.py:
`pixel_grid` :9-19, `sample` :201-208, and the depth-based warp helpers of the offline stages (`depth_to_points`
:130-139, `calibrate_scale` :142-176, `warping_field` :179-200, `warp_image` :213-227).  On the hot path the same
projection algebra is folded into the fused loss kernel."""
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


def _frames(depths, intrinsics, extrinsics=None):
    depths = depths.float().contiguous()
    N, one, H, W = depths.shape
    if one != 1 or tuple(intrinsics.shape) != (N, 4) or (extrinsics is not None and tuple(extrinsics.shape) != (N, 3, 4)):
        raise ValueError("expected depths (N,1,H,W), intrinsics (N,4) = fx,fy,cx,cy and extrinsics (N,3,4) = [R|t]")
    return depths, intrinsics.float().contiguous(), (extrinsics.float().contiguous() if extrinsics is not None else None), N, H, W


def depth_to_points(depths, intrinsics):
    """Camera-space points (N,3,H,W) of depths (N,1,H,W) (camera looks along -z)."""
    depths, intrinsics, _, N, H, W = _frames(depths, intrinsics)
    out = torch.empty(N, 3, H, W, dtype=torch.float32, device=depths.device)
    rc = _native.lib().cd_depth_to_points(_native.dev_ptr(depths, "depths"), _native.dev_ptr(intrinsics, "intrinsics"), N, H, W,
                                         out.data_ptr(), None, _native.stream_ptr(depths.device))
    _native.check(rc, "cd_depth_to_points")
    return out


def calibrate_scale(extrinsics, intrinsics, depths):
    """Global scale that reconciles the two scene centres with the camera baseline (2 frames):
    -dt.dmu / dt.dt with mu_i = R_i mean(points_i), dt = t_0 - t_1.  Returns a 0-d fp64 tensor on the device."""
    depths, intrinsics, extrinsics, N, H, W = _frames(depths, intrinsics, extrinsics)
    if N != 2:
        raise ValueError("calibrate_scale takes exactly two frames (as the reference asserts)")
    sums = torch.empty(N, 3, dtype=torch.float64, device=depths.device)
    rc = _native.lib().cd_depth_to_points(_native.dev_ptr(depths, "depths"), _native.dev_ptr(intrinsics, "intrinsics"), N, H, W,
                                         None, sums.data_ptr(), _native.stream_ptr(depths.device))
    _native.check(rc, "cd_depth_to_points")
    E = extrinsics.double()
    mus = torch.bmm(E[:, :, :3], (sums / (H * W)).unsqueeze(-1)).squeeze(-1)     # 2 x (3x3 @ 3): host-sized plumbing
    dmu, dt = mus[0] - mus[1], E[0, :, 3] - E[1, :, 3]
    return -dt.dot(dmu) / dt.dot(dt)


def _warp(images, depths, extrinsics, intrinsics, tgt_ids, want_uv, want_img):
    depths, intrinsics, extrinsics, N, H, W = _frames(depths, intrinsics, extrinsics)
    ids = torch.as_tensor(list(tgt_ids) if not torch.is_tensor(tgt_ids) else tgt_ids, dtype=torch.int32, device=depths.device).reshape(-1)
    if ids.numel() != N or int(ids.min()) < 0 or int(ids.max()) >= N:
        raise ValueError("tgt_ids must name one target frame in [0, N) per frame")
    C = 0
    if want_img:
        images = images.float().contiguous()
        if images.shape[0] != N or tuple(images.shape[2:]) != (H, W):
            raise ValueError("images must be (N, C, H, W) like the depths")
        C = images.shape[1]
    uv = torch.empty(N, 2, H, W, dtype=torch.float32, device=depths.device) if want_uv else None
    out = torch.empty(N, C, H, W, dtype=torch.float32, device=depths.device) if want_img else None
    rc = _native.lib().cd_warp_image(_native.dev_ptr(images, "images") if want_img else None, _native.dev_ptr(depths, "depths"),
                                    _native.dev_ptr(intrinsics, "intrinsics"), _native.dev_ptr(extrinsics, "extrinsics"),
                                    ids.data_ptr(), N, C, H, W, uv.data_ptr() if want_uv else None,
                                    out.data_ptr() if want_img else None, _native.stream_ptr(depths.device))
    _native.check(rc, "cd_warp_image")
    return uv, out


def warping_field(extrinsics, intrinsics, depths, tgt_ids):
    """uv (N,2,H,W): sampling frame tgt_ids[i] at uv[i] reproduces frame i."""
    return _warp(None, depths, extrinsics, intrinsics, tgt_ids, True, False)[0]


def warp_image(images, depths, extrinsics, intrinsics, tgt_ids):
    """images[tgt_ids[i]] warped into frame i by the depth of frame i and the two poses -- one fused launch."""
    return _warp(images, depths, extrinsics, intrinsics, tgt_ids, False, True)[1]
