"""Scale calibration, the part that feeds the fine-tuning path: per-frame median scales of the initial depth maps against COLMAP's
dense depth, `scales.csv`, the scaled depth maps and `metadata_scaled.npz` (camera translations divided by the mean scale) -- the
file `loaders/video_dataset.py` / `PairStore.from_directory` read.

Mirrors /root/reference/scale_calibration.py:228-319 (`calibrate_scale`, from "Compute per-frame scales" on): same file names, same
skip rules (a frame without a converted COLMAP map is skipped; a frame whose finite COLMAP pixels are fewer than
--dense_pixel_ratio of the image is invalid), same numbers: the medians come from ONE launch of `cd_frame_median_scales`
(csrc/scale.hip: exact selection, bit for bit np.median), the rest is host bookkeeping on a handful of floats.
What is NOT here: running COLMAP and converting its depth maps (:84-226, out of scope: SURVEY.md section 2), the PNG visualisations.
`ScaleCalibrationParams` carries the reference's two flags (:25-34).
"""
from __future__ import annotations

import os
from os.path import join as pjoin

import numpy as np
import torch

from . import _native
from .utils import image_io


class ScaleCalibrationParams:
    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--dense_frame_ratio", type=float, default=0.95,
                            help="threshold on percentage of successully computed dense depth frames.")
        parser.add_argument("--dense_pixel_ratio", type=float, default=0.3,
                            help="ratio of valid dense depth pixels for that frame to valid")
        return parser


def nearest_resize(img: np.ndarray, shape) -> np.ndarray:
    """cv2.resize(img, shape[::-1], interpolation=cv2.INTER_NEAREST) (:264-267): source index = min(floor(dst * src / dst_size), src - 1),
    computed in double like OpenCV's resizeNN."""
    H, W = shape
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img
    ys = np.minimum(np.floor(np.arange(H) * (1.0 / (H / float(h)))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W) * (1.0 / (W / float(w)))).astype(np.int64), w - 1)
    return img[ys][:, xs]


def frame_median_scales(inv_src: torch.Tensor, inv_cmp: torch.Tensor, want_scaled: bool = True):
    """inv_src, inv_cmp (N,H,W) float32 on the HIP device -> (scales (N,) float32, n_valid (N,) int32, inv_src / scale (N,H,W) or None)."""
    inv_src, inv_cmp = inv_src.float().contiguous(), inv_cmp.float().contiguous()
    if inv_src.shape != inv_cmp.shape or inv_src.dim() != 3:
        raise ValueError(f"expected two (N,H,W) tensors, got {tuple(inv_src.shape)} and {tuple(inv_cmp.shape)}")
    N, H, W = inv_src.shape
    scales = torch.empty(N, dtype=torch.float32, device=inv_src.device)
    n_valid = torch.empty(N, dtype=torch.int32, device=inv_src.device)
    scaled = torch.empty_like(inv_src) if want_scaled else None
    rc = _native.lib().cd_frame_median_scales(_native.dev_ptr(inv_src, "inv_src"), _native.dev_ptr(inv_cmp, "inv_cmp"), N, H, W,
                                              scales.data_ptr(), n_valid.data_ptr(), scaled.data_ptr() if want_scaled else None,
                                              _native.stream_ptr(inv_src.device))
    _native.check(rc, "cd_frame_median_scales")
    return scales, n_valid, scaled


def compute_frame_scales(frames, src_depth_fmt: str, converted_depth_fmt: str, scaled_depth_fmt: str, dense_pixel_ratio: float = 0.3,
                         device=None, chunk: int = 64):
    """:253-278.  Returns {frame: float(scale)} of the valid frames and writes their scaled inverse depth maps."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    os.makedirs(os.path.dirname(scaled_depth_fmt), exist_ok=True)
    present = [i for i in frames if os.path.isfile(converted_depth_fmt.format(i))]       # (:257-261: missing maps are skipped)
    scales_map = {}
    for s0 in range(0, len(present), chunk):
        ids = present[s0:s0 + chunk]
        src = [image_io.load_raw_float32_image(src_depth_fmt.format(i)) for i in ids]
        cmp_ = [nearest_resize(image_io.load_raw_float32_image(converted_depth_fmt.format(i)), src[k].shape[:2]) for k, i in enumerate(ids)]
        inv_src = torch.as_tensor(np.stack(src)).to(device)
        inv_cmp = torch.as_tensor(np.stack(cmp_)).to(device)
        scales, n_valid, scaled = frame_median_scales(inv_src, inv_cmp)
        scales, n_valid, scaled = scales.cpu().numpy(), n_valid.cpu().numpy(), scaled.cpu().numpy()
        size = inv_src.shape[1] * inv_src.shape[2]
        for k, i in enumerate(ids):
            if n_valid[k] / size < dense_pixel_ratio:        # not enough valid pixels: the frame is invalid (:270-272)
                continue
            print(f"Scale[{i}]: median={scales[k]}")
            scales_map[i] = float(scales[k])
            image_io.save_raw_float32_image(scaled_depth_fmt.format(i), scaled[k])
    return scales_map


def write_scales_csv(scales_file: str, scales_map) -> np.ndarray:
    """:286-290: (M, 2) rows (frame, scale), sorted by frame, float64, comma separated."""
    xs = sorted(scales_map.keys())
    ys = [scales_map[x] for x in xs]
    src_to_colmap_scales = np.stack((np.array(xs), np.array(ys)), axis=-1)
    np.savetxt(scales_file, src_to_colmap_scales, delimiter=",")
    return src_to_colmap_scales


def write_scaled_metadata(src_meta_file: str, scaled_meta_file: str, src_to_colmap_scales: np.ndarray):
    """:296-311: translations divided by the mean scale; keys intrinsics / extrinsics / scales."""
    scales = src_to_colmap_scales[:, 1]
    mean_scale = scales.mean()
    print(f"[scales] mean={mean_scale}, std={np.std(scales)}")
    with np.load(src_meta_file) as meta_colmap:
        intrinsics = meta_colmap["intrinsics"]
        extrinsics = meta_colmap["extrinsics"]
    extrinsics[..., -1] /= mean_scale
    np.savez(scaled_meta_file, intrinsics=intrinsics, extrinsics=extrinsics, scales=src_to_colmap_scales)
    return intrinsics, extrinsics


def calibrate_scale(path: str, out_dir: str, frames, model_type: str = "mc", dense_frame_ratio: float = 0.95, dense_pixel_ratio: float = 0.3):
    """The stage from "Compute per-frame scales" on (:228-311) with the reference's directory layout:
        <path>/depth_<model_type>/depth/frame_%06d.raw        initial inverse depth (the depth model's export)
        <path>/depth_colmap_dense/depth/frame_%06d.raw        COLMAP's dense depth as inverse depth, NaN = no value (:190-218)
        <out_dir>/metadata.npz                                 COLMAP cameras (:180-183)
    ->  <out_dir>/depth_scaled_by_colmap_dense/depth/frame_%06d.raw, <out_dir>/scales.csv, <out_dir>/metadata_scaled.npz.
    Returns the set of valid frames (:292)."""
    converted_depth_fmt = pjoin(path, "depth_colmap_dense", "depth", "frame_{:06d}.raw")
    scaled_depth_fmt = pjoin(out_dir, "depth_scaled_by_colmap_dense", "depth", "frame_{:06d}.raw")
    scales_file = pjoin(out_dir, "scales.csv")
    src_depth_fmt = pjoin(path, f"depth_{model_type}", "depth", "frame_{:06d}.raw")
    src_meta_file = pjoin(out_dir, "metadata.npz")
    scaled_meta_file = pjoin(out_dir, "metadata_scaled.npz")
    frames = list(frames)
    if os.path.isfile(scales_file) and all(os.path.isfile(scaled_depth_fmt.format(i)) for i in frames
                                           if os.path.isfile(converted_depth_fmt.format(i))):
        src_to_colmap_scales = np.loadtxt(scales_file, delimiter=",").reshape(-1, 2)
        assert src_to_colmap_scales.shape[0] >= len(frames) * dense_frame_ratio and src_to_colmap_scales.shape[1] == 2, \
            f"scales shape is {src_to_colmap_scales.shape} does not match ({len(frames)}, 2) with threshold {dense_frame_ratio}"
        print("Existing scales file loaded.")
    else:
        scales_map = compute_frame_scales(frames, src_depth_fmt, converted_depth_fmt, scaled_depth_fmt, dense_pixel_ratio)
        src_to_colmap_scales = write_scales_csv(scales_file, scales_map)
    valid_frames = {int(s) for s in src_to_colmap_scales[:, 0]}
    if os.path.isfile(scaled_meta_file):
        print("Scaled metadata file exists.")
    else:
        write_scaled_metadata(src_meta_file, scaled_meta_file, src_to_colmap_scales)
    return valid_frames
