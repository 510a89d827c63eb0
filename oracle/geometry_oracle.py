"""TEST INFRASTRUCTURE ONLY -- numpy (fp64) restatement of the reference's depth-based warp helpers
(/root/reference/utils/geometry.py):
    depth_to_points   :130-139   (pixel_grid :9-19, pixels_to_rays :38-61, pixels_to_points :86-100)
    calibrate_scale   :142-176
    warping_field     :179-200   (reproject_points :103-128, project :64-83)
    warp_image        :213-227   (sample :201-208 -> oracle.sample, the C restatement already pinned)
Only tests/ may import this.  Pinned by tests/golden/warp_*.npz (oracle/gen_golden_warp.py runs the reference itself).
Conventions: intrinsics row = (fx, fy, cx, cy); extrinsics = [R | t], x_world = R p + t; the camera looks
along -z; pixel y grows downwards, camera y upwards.
"""
import numpy as np


def depth_to_points(depths, intrinsics):
    """depths (B,1,H,W), intrinsics (B,4) -> camera-space points (B,3,H,W)"""
    B, _, H, W = depths.shape
    x = np.arange(W, dtype=np.float64)[None, None, :]
    y = np.arange(H, dtype=np.float64)[None, :, None]
    fx, fy, cx, cy = (intrinsics[:, i].astype(np.float64)[:, None, None] for i in range(4))
    d = depths[:, 0].astype(np.float64)
    rx = (x - cx) / fx + 0 * y
    ry = -(y - cy) / fy + 0 * x
    return np.stack((rx * d, ry * d, -d), axis=1)


def calibrate_scale(extrinsics, intrinsics, depths):
    """scale = -dt.dmu / dt.dt with mu_i = mean(R_i p_cam), dt = t_0 - t_1   [:142-176]"""
    pts = depth_to_points(depths, intrinsics)
    B = pts.shape[0]
    assert B == 2
    R, t = extrinsics[..., :3].astype(np.float64), extrinsics[..., 3].astype(np.float64)
    mus = np.einsum("bij,bjn->bin", R, pts.reshape(B, 3, -1)).mean(-1)
    dmu, dt = mus[0] - mus[1], t[0] - t[1]
    return -dt.dot(dmu) / dt.dot(dt)


def warping_field(extrinsics, intrinsics, depths, tgt_ids):
    """uv (N,2,H,W): sampling frame tgt_ids[i] at uv[i] reproduces frame i   [:179-200]"""
    pts = depth_to_points(depths, intrinsics)
    N, _, H, W = pts.shape
    R, t = extrinsics[..., :3].astype(np.float64), extrinsics[..., 3].astype(np.float64)
    Rt, tt, it = R[tgt_ids], t[tgt_ids], intrinsics[tgt_ids].astype(np.float64)
    world = np.einsum("bij,bjn->bin", R, pts.reshape(N, 3, -1)) + t[:, :, None]
    cam_t = np.einsum("bji,bjn->bin", Rt, world - tt[:, :, None]).reshape(N, 3, H, W)
    rays = cam_t / -cam_t[:, 2:3]
    fx, fy, cx, cy = (it[:, i][:, None, None] for i in range(4))
    u = rays[:, 0] * fx + cx
    v = -(rays[:, 1] * fy) + cy
    return np.stack((u, v), axis=1)


def warp_image(images, depths, extrinsics, intrinsics, tgt_ids):
    """images[tgt_ids[i]] warped to frame i   [:213-227]"""
    from oracle import oracle
    uv = warping_field(extrinsics, intrinsics, depths, tgt_ids)
    return oracle.sample(images[tgt_ids].astype(np.float64), uv, dtype=np.float64), uv
