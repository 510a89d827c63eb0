"""Frame-pair dataset over the reference's on-disk layout (host side).

Same classes, file layout and per-item output as /root/reference/loaders/video_dataset.py:80-242:

    <path>/color_down/frame_%06d.raw      BGR float [0,1]  (read and swapped to RGB, :56)
    <path>/flow/flow_%06d_%06d.raw        (H,W,2) fp32 pixels, ref -> tgt
    <path>/mask/mask_%06d_%06d.png        8-bit, > 0 = valid
    <meta_file> (metadata_scaled.npz)     intrinsics (N,4) fx,fy,cx,cy ; extrinsics (N,3,4) [R|t]
    <path>/flow_list.json                 [[i,j], ...] both directions -> unique ordered pairs

`VideoDataset[i]` -> (images (2,3,H,W), metadata) with metadata["geometry_consistency"] =
{"indices": (2,), "flows": [(2,H,W)]*2, "masks": [(1,H,W)]*2}.  This is the compatibility
path (CPU tensors, DataLoader workers); the fast path keeps everything in HBM
(consistent_depth_amd/loaders/pair_store.py).
"""
from __future__ import annotations

import json
import os
from os.path import join as pjoin

import numpy as np
import torch
import torch.utils.data as data

from ..utils import frame_sampling as sampling
from ..utils import image_io

_dtype = torch.float32


def load_color(path: str, channels_first: bool = True) -> torch.Tensor:
    """RGB in [0,1]; .raw files hold BGR."""
    if path.endswith(".raw"):
        im = image_io.load_raw_float32_image(path)
        if im.ndim == 3:
            im = im[..., ::-1]
    else:  # PNG fallback of the reference: value/255, channel order as stored
        from PIL import Image
        with Image.open(path) as f:
            im = np.asarray(f, dtype=np.float32)[..., ::-1] / 255.0  # cv2.imread order is BGR
    im = im.reshape(im.shape[:2] + (-1,))
    if channels_first:
        im = im.transpose(2, 0, 1)
    return torch.tensor(np.ascontiguousarray(im), dtype=_dtype)


def load_flow(path: str, channels_first: bool = True) -> torch.Tensor:
    f = image_io.load_raw_float32_image(path)
    if f.ndim != 3 or f.shape[-1] != 2:
        raise ValueError(f"{path}: flow must have 2 channels, got shape {f.shape}")
    return torch.tensor(np.ascontiguousarray(f.transpose(2, 0, 1) if channels_first else f), dtype=_dtype)


def load_mask(path: str, channels_first: bool = True) -> torch.Tensor:
    m = image_io.load_mask_png(path).astype(np.float32)
    return torch.tensor(m[None] if channels_first else m[..., None], dtype=_dtype)


def read_pair_list(path: str, flow_dir: str = None):
    """Unique ordered (i < j) pairs: flow_list.json if present, else parsed from the flow file names."""
    fn = pjoin(path, "flow_list.json")
    if os.path.isfile(fn):
        with open(fn) as f:
            pairs = json.load(f)
    else:
        flow_dir = flow_dir or pjoin(path, "flow")
        pairs = []
        for name in os.listdir(flow_dir):
            stem, ext = os.path.splitext(name)
            if ext == ".raw":
                pairs.append([int(s) for s in stem.split("_")[-2:]])
    return list(sampling.SamplePairs.to_one_way(pairs))


class VideoDataset(data.Dataset):
    def __init__(self, path: str, meta_file: str = None):
        self.color_fmt = pjoin(path, "color_down", "frame_{:06d}.raw")
        if not os.path.isfile(self.color_fmt.format(0)):
            self.color_fmt = pjoin(path, "color_down", "frame_{:06d}.png")
        self.mask_fmt = pjoin(path, "mask", "mask_{:06d}_{:06d}.png")
        self.flow_fmt = pjoin(path, "flow", "flow_{:06d}_{:06d}.raw")
        if meta_file is not None:
            with np.load(meta_file) as meta:
                self.extrinsics = torch.tensor(meta["extrinsics"], dtype=_dtype)
                self.intrinsics = torch.tensor(meta["intrinsics"], dtype=_dtype)
            if self.extrinsics.shape[0] != self.intrinsics.shape[0]:
                raise ValueError(f"#extrinsics({self.extrinsics.shape[0]}) != #intrinsics({self.intrinsics.shape[0]})")
        self.flow_indices = read_pair_list(path)

    def __len__(self):
        return len(self.flow_indices)

    def __getitem__(self, index: int):
        i, j = self.flow_indices[index]
        images = torch.stack([load_color(self.color_fmt.format(k)) for k in (i, j)], 0)
        metadata = {
            "extrinsics": torch.stack([self.extrinsics[i], self.extrinsics[j]], 0),
            "intrinsics": torch.stack([self.intrinsics[i], self.intrinsics[j]], 0),
            "geometry_consistency": {
                "indices": torch.tensor([i, j]),
                "flows": [load_flow(self.flow_fmt.format(a, b)) for a, b in ((i, j), (j, i))],
                "masks": [load_mask(self.mask_fmt.format(a, b)) for a, b in ((i, j), (j, i))],
            },
        }
        return images, metadata


class VideoFrameDataset(data.Dataset):
    """Single frames for depth export: item -> (image (3,H,W), {"frame_id": id})."""

    def __init__(self, color_fmt, frames=None):
        self.color_fmt = color_fmt
        self.frames = frames if frames is not None else range(len(os.listdir(os.path.dirname(color_fmt))))

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, index):
        frame_id = self.frames[index]
        return load_color(self.color_fmt.format(frame_id)), {"frame_id": frame_id}
