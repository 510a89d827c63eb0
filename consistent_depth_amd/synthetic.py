"""Seeded synthetic frame-pair data in the reference's batch layout.

There is no network (no ayush clip, no COLMAP, no FlowNet2), so every config of
BASELINE.json is driven by synthetic videos built here (recipe: SURVEY.md section 8d):

* one pinhole camera for the whole clip, fx = fy = 1671.770118 * W / 1080
  (reference README.md:52 scaled like utils/load_colmap.py:132-135), cx = W/2, cy = H/2;
* a smooth camera path (small yaw/pitch, random-walk translation), extrinsics = [R|t]
  with x_world = R p + t and an OpenGL-style camera (looks down -z, y up) as produced by
  the reference's calibration stage (utils/load_colmap.py:31,139-158);
* a smooth positive ground-truth depth field per frame in [0.5, 4];
* flow = exact reprojection flow of that depth + N(0, noise_px) so the loss has a
  meaningful minimum, mask = Bernoulli(keep) AND in-bounds, stored as float {0,1}
  (loaders/video_dataset.py:71-77), colours U[0,1).

Everything is numpy on the host; callers move tensors to the GPU.
The layout of one batch mirrors what loaders/video_dataset.py:131-207 + default collate
hand to the loss (loss/consistency_loss.py:98-127):
    depth (B,2,H,W)  flows [ (B,2,H,W) ]*2  masks [ (B,1,H,W) ]*2
    intrinsics (B,2,4)  extrinsics (B,2,3,4)
"""
from __future__ import annotations

import numpy as np

AYUSH_FOCAL_1080 = 1671.770118  # reference README.md:52


def clip_intrinsics(H: int, W: int) -> np.ndarray:
    f = AYUSH_FOCAL_1080 * W / 1080.0
    return np.array([f, f, W / 2.0, H / 2.0], dtype=np.float64)


def _rot_yx(yaw: float, pitch: float) -> np.ndarray:
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return ry @ rx


def camera_path(n_frames: int, rng: np.random.Generator, step: float = 0.02,
                max_angle: float = 0.3) -> np.ndarray:
    """(N,3,4) [R|t], camera-to-world."""
    ext = np.zeros((n_frames, 3, 4))
    t = np.zeros(3)
    # smooth angles: integrated, rescaled to stay below max_angle over the clip
    a = np.cumsum(rng.normal(0, 1, (n_frames, 2)), axis=0)
    a *= max_angle / max(1e-9, np.abs(a).max())
    for i in range(n_frames):
        ext[i, :, :3] = _rot_yx(a[i, 0], a[i, 1])
        ext[i, :, 3] = t
        t = t + rng.normal(0, step, 3)
    return ext


def smooth_field(H: int, W: int, rng: np.random.Generator, lo: float, hi: float,
                 n_waves: int = 4) -> np.ndarray:
    """Smooth (H,W) field in [lo, hi]: a few random low-frequency sinusoids."""
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    f = np.zeros((H, W))
    for _ in range(n_waves):
        kx, ky = rng.uniform(0.5, 3.0, 2) * 2 * np.pi * rng.choice([-1, 1], 2)
        f += rng.uniform(0.3, 1.0) * np.sin(kx * xx + ky * yy + rng.uniform(0, 2 * np.pi))
    f = (f - f.min()) / max(1e-12, f.max() - f.min())
    return lo + (hi - lo) * f


def reprojection_flow(depth_ref: np.ndarray, intr_ref, extr_ref, intr_tgt, extr_tgt) -> np.ndarray:
    """Exact ref->tgt flow (2,H,W) of a depth map: unproject, move, project, minus grid."""
    H, W = depth_ref.shape
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    fx, fy, cx, cy = intr_ref
    ray = np.stack([(x - cx) / fx, -(y - cy) / fy, -np.ones_like(x)], 0)  # looks down -z
    p = ray * depth_ref[None]
    R_r, t_r = extr_ref[:, :3], extr_ref[:, 3]
    R_t, t_t = extr_tgt[:, :3], extr_tgt[:, 3]
    world = np.einsum("ij,jhw->ihw", R_r, p) + t_r[:, None, None]
    q = np.einsum("ji,jhw->ihw", R_t, world - t_t[:, None, None])
    fxt, fyt, cxt, cyt = intr_tgt
    px = fxt * q[0] / (-q[2]) + cxt
    py = -(fyt * q[1] / (-q[2])) + cyt
    return np.stack([px - x, py - y], 0)


def make_pair_batch(B: int, H: int, W: int, seed: int = 0, noise_px: float = 0.5,
                    mask_keep: float = 0.7, depth_jitter: float = 0.05,
                    frame_gap: int = 4, dtype=np.float32) -> dict:
    """One batch of B independent frame pairs (each from its own short camera path)."""
    rng = np.random.default_rng(seed)
    K = clip_intrinsics(H, W)
    depth = np.zeros((B, 2, H, W))
    flows = [np.zeros((B, 2, H, W)), np.zeros((B, 2, H, W))]
    masks = [np.zeros((B, 1, H, W)), np.zeros((B, 1, H, W))]
    intr = np.tile(K, (B, 2, 1))
    extr = np.zeros((B, 2, 3, 4))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        path = camera_path(frame_gap + 1, rng)
        extr[b, 0], extr[b, 1] = path[0], path[frame_gap]
        gt = [smooth_field(H, W, rng, 0.5, 4.0) for _ in range(2)]
        for k in range(2):
            f = reprojection_flow(gt[k], K, extr[b, k], K, extr[b, 1 - k])
            f = f + rng.normal(0, noise_px, f.shape)
            flows[k][b] = f
            inb = ((xx + f[0] >= 0) & (xx + f[0] <= W - 1) & (yy + f[1] >= 0) & (yy + f[1] <= H - 1))
            masks[k][b, 0] = (rng.random((H, W)) < mask_keep) & inb
            # the network's current estimate: GT times a smooth multiplicative error
            depth[b, k] = gt[k] * np.exp(smooth_field(H, W, rng, -depth_jitter, depth_jitter))
    return {
        "depth": depth.astype(dtype),
        "flows": [f.astype(dtype) for f in flows],
        "masks": [m.astype(dtype) for m in masks],
        "intrinsics": intr.astype(dtype),
        "extrinsics": extr.astype(dtype),
    }


# ---------------------------------------------------------------------------------------------
# Geometrically CONSISTENT scenes: one static surface seen by every camera, so forward and
# backward flows are mutual inverses wherever the mask is 1 -- which is what the reference's
# flow-consistency masks certify on real video (flow.py:199-228).  `make_pair_batch` above gives
# every frame its own unrelated depth field (flows of a pair disagree by tens of pixels); it is
# kept as the adversarial generator for the parity tests.
# ---------------------------------------------------------------------------------------------
class Surface:
    """Static scene: height field z_world = -D(x_world, y_world), D smooth in [lo, hi]."""

    def __init__(self, rng: np.random.Generator, lo: float = 1.5, hi: float = 4.0, n_waves: int = 4):
        self.k = rng.uniform(0.4, 1.2, (n_waves, 2)) * rng.choice([-1.0, 1.0], (n_waves, 2))
        self.phase = rng.uniform(0, 2 * np.pi, n_waves)
        a = rng.uniform(0.3, 1.0, n_waves)
        self.amp = a / a.sum() * (hi - lo) / 2.0
        self.mid = (hi + lo) / 2.0

    def depth_below(self, x, y):
        d = np.full_like(x, self.mid)
        for (kx, ky), ph, a in zip(self.k, self.phase, self.amp):
            d = d + a * np.sin(kx * x + ky * y + ph)
        return d


def render_depth(surface: Surface, intr, extr, H: int, W: int, iters: int = 12) -> np.ndarray:
    """Depth map (distance along the camera's -z) of the surface: fixed-point ray/height-field
    intersection (the surface is gentle, the iteration contracts)."""
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    fx, fy, cx, cy = intr
    ray = np.stack([(x - cx) / fx, -(y - cy) / fy, -np.ones_like(x)], 0)
    R, t = extr[:, :3], extr[:, 3]
    d = np.einsum("ij,jhw->ihw", R, ray)  # world direction, d[2] < 0
    s = np.full((H, W), surface.mid)
    for _ in range(iters):
        px, py = t[0] + s * d[0], t[1] + s * d[1]
        s = (surface.depth_below(px, py) + t[2]) / (-d[2])
    return s


def make_scene_batch(B: int, H: int, W: int, seed: int = 0, noise_px: float = 0.25, mask_keep: float = 0.7,
                     depth_jitter: float = 0.05, frame_gap: int = 4, dtype=np.float32) -> dict:
    """Like make_pair_batch, but both frames of a pair look at ONE surface (consistent flows)."""
    rng = np.random.default_rng(seed)
    K = clip_intrinsics(H, W)
    depth = np.zeros((B, 2, H, W))
    flows = [np.zeros((B, 2, H, W)), np.zeros((B, 2, H, W))]
    masks = [np.zeros((B, 1, H, W)), np.zeros((B, 1, H, W))]
    extr = np.zeros((B, 2, 3, 4))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for b in range(B):
        surf = Surface(rng)
        path = camera_path(frame_gap + 1, rng, max_angle=0.1)
        extr[b, 0], extr[b, 1] = path[0], path[frame_gap]
        gt = [render_depth(surf, K, extr[b, k], H, W) for k in range(2)]
        for k in range(2):
            f = reprojection_flow(gt[k], K, extr[b, k], K, extr[b, 1 - k]) + rng.normal(0, noise_px, (2, H, W))
            flows[k][b] = f
            inb = ((xx + f[0] >= 0) & (xx + f[0] <= W - 1) & (yy + f[1] >= 0) & (yy + f[1] <= H - 1))
            masks[k][b, 0] = (rng.random((H, W)) < mask_keep) & inb
            depth[b, k] = gt[k] * np.exp(smooth_field(H, W, rng, -depth_jitter, depth_jitter))
    return {"depth": depth.astype(dtype), "flows": [f.astype(dtype) for f in flows],
            "masks": [m.astype(dtype) for m in masks], "intrinsics": np.tile(K, (B, 2, 1)).astype(dtype),
            "extrinsics": extr.astype(dtype)}


def make_video(n_frames: int, H: int, W: int, seed: int = 0, step: float = 0.01, max_angle: float = 0.15):
    """A whole synthetic clip: colours (N,3,H,W) U[0,1), GT depth (N,H,W), cameras.  `step` / `max_angle` bound the camera motion
    (small values keep every pixel of a frame inside the other frame of its pairs: a clip whose masks can cover everything)."""
    rng = np.random.default_rng(seed)
    K = clip_intrinsics(H, W)
    extr = camera_path(n_frames, rng, step=step, max_angle=max_angle)
    surf = Surface(rng)
    depth = np.stack([render_depth(surf, K, extr[i], H, W) for i in range(n_frames)])
    color = rng.random((n_frames, 3, H, W), dtype=np.float32)
    return {"color": color, "gt_depth": depth, "intrinsics": np.tile(K, (n_frames, 1)),
            "extrinsics": extr}


def video_pair_data(video: dict, i: int, j: int, rng: np.random.Generator,
                    noise_px: float = 0.5, mask_keep: float = 0.7):
    """flow/mask (both directions) for frames (i, j) of a `make_video` clip."""
    K, extr, gt = video["intrinsics"], video["extrinsics"], video["gt_depth"]
    H, W = gt.shape[1:]
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    out = []
    for a, b in ((i, j), (j, i)):
        f = reprojection_flow(gt[a], K[a], extr[a], K[b], extr[b])
        f = f + rng.normal(0, noise_px, f.shape)
        inb = ((xx + f[0] >= 0) & (xx + f[0] <= W - 1) & (yy + f[1] >= 0) & (yy + f[1] <= H - 1))
        m = ((rng.random((H, W)) < mask_keep) & inb)
        out.append((f.astype(np.float32), m.astype(np.float32)[None]))
    return out
