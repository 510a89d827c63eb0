"""Device-resident frame-pair store.

The reference reads ~3.5 MB of files per pair per step through 4 DataLoader worker processes
(loaders/video_dataset.py:131-207, depth_fine_tuning.py:205-218).  At MI355X step rates that
loader is the bottleneck, and 288 GB of HBM holds any realistic clip outright (244 frames at
384x224: 0.25 GB of colour + 715 pairs x 2.06 MB of flow/mask = 1.7 GB; 1000 frames / 2979 pairs:
7.2 GB).  So the whole dataset is uploaded ONCE and mini-batches are gathered on the GPU:

    colour  (F,3,H,W) fp32      flows (P,2,2,H,W) fp32 [pair, direction, (dx,dy)]
    masks   (P,2,1,H,W) uint8   one byte per pixel like the reference's mask PNGs (video_dataset.py:71-77: `> 0` -> float);
                                widened to the loss kernels' fp32 {0,1} by the gather: 34 H W bytes per pair resident
                                instead of 40
    mask_sums (P,2) fp32, tile_windows (P, bytes) uint8   dataset constants of the loss kernels (normalisers, source windows
                                and row-sweep plans), computed once
    intrinsics (F,4), extrinsics (F,3,4), pair_frames (P,2) int64 (indices into F)

`batch(pair_ids)` returns exactly what default_collate + to_device hand the reference's loop -- ONE HIP launch
(cd_gather_pairs, csrc/store.hip) instead of file reads, numpy conversions and H2D copies; `gather_into` writes straight
into the static input buffers of a captured step graph.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _native
from ..loss.consistency_loss import mask_sums as _mask_sums, tile_windows as _tile_windows


class _StoreDesc(ctypes.Structure):     # cd_pair_store of include/consistent_depth_amd.h
    _fields_ = [(n, ctypes.c_void_p) for n in ("color", "flows", "masks", "intrinsics", "extrinsics", "pair_frames", "frame_ids",
                                               "mask_sums", "plans")] + \
               [("plan_bytes", ctypes.c_int64)] + [(n, ctypes.c_int32) for n in ("F", "P", "H", "W", "mask_u8", "reserved")]


class _BatchDesc(ctypes.Structure):     # cd_pair_batch
    _fields_ = [(n, ctypes.c_void_p) for n in ("images", "flow_fwd", "flow_bwd", "mask_fwd", "mask_bwd", "intrinsics", "extrinsics",
                                               "indices", "mask_sums", "plans")]


class PairStore:
    def __init__(self, color, flows, masks, intrinsics, extrinsics, pair_frames, frame_ids=None, device=None):
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type == "cuda" and dev.index is None:      # "cuda" -> "cuda:<current>": tensors report an indexed device
            dev = torch.device("cuda", torch.cuda.current_device())
        # (arrays from the host, or tensors that already live on the device: synthetic_device)
        f32 = lambda a: (a.to(dev, torch.float32).contiguous() if torch.is_tensor(a)  # noqa: E731
                         else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev))
        self.device = dev
        self.color = f32(color)
        self.flows = f32(flows)
        self.masks = ((masks.to(dev) if torch.is_tensor(masks) else torch.as_tensor(np.ascontiguousarray(masks))) > 0).to(torch.uint8).to(dev).contiguous()
        self.intrinsics = f32(intrinsics)
        self.extrinsics = f32(extrinsics)
        self.pair_frames = torch.as_tensor(np.asarray(pair_frames), dtype=torch.int64).to(dev)
        # original frame numbers (for file names / logs); row r of colour is frame frame_ids[r]
        self.frame_ids = list(frame_ids) if frame_ids is not None else list(range(self.color.shape[0]))
        self._frame_ids_dev = torch.as_tensor(self.frame_ids, dtype=torch.int64).to(dev)
        P = self.flows.shape[0]
        assert self.masks.shape[0] == P and self.pair_frames.shape == (P, 2)
        self._refresh_constants()

    def _refresh_constants(self):
        """Dataset constants derived from the masks and flows: the loss normalisers and the gradient kernel's windows."""
        P, dev = self.flows.shape[0], self.device
        self.mask_sums = torch.empty(P, 2, dtype=torch.float32, device=dev)
        wins = []
        for s in range(0, P, 256):
            fl = [self.flows[s:s + 256, 0].contiguous(), self.flows[s:s + 256, 1].contiguous()]
            mk = [self.masks[s:s + 256, 0].float().contiguous(), self.masks[s:s + 256, 1].float().contiguous()]
            self.mask_sums[s:s + 256] = _mask_sums(mk[0], mk[1])
            wins.append(_tile_windows(fl, mk))
        self.tile_windows = torch.cat(wins, 0).contiguous()  # (P, bytes_per_pair) uint8
        self._desc = None

    def rebuild_masks(self, flow_thresh: float = 1.0, color_thresh: float = 1.0):
        """Recompute the flow-consistency masks of every pair ON THE DEVICE from the resident flows and colours -- the
        reference's offline `mask_valid_correspondences` stage (flow.py:199-228 -> utils/consistency.py), bit-identical
        masks -- and refresh the constants that depend on them.  For datasets that ship flows without mask PNGs, or to
        try other thresholds without touching the disk."""
        from ..utils import consistency
        for s in range(0, len(self), 256):
            pf = self.pair_frames[s:s + 256]
            m0, m1 = consistency.consistent_flow_masks_batch(
                self.flows[s:s + 256, 0].contiguous(), self.flows[s:s + 256, 1].contiguous(),
                self.color[pf[:, 0]], self.color[pf[:, 1]], flow_thresh, color_thresh)
            self.masks[s:s + 256, 0] = (m0 > 0).to(torch.uint8)
            self.masks[s:s + 256, 1] = (m1 > 0).to(torch.uint8)
        self._refresh_constants()
        return self

    @classmethod
    def synthetic_device(cls, n_frames: int, H: int, W: int, flow_ops=("hierarchical2",), seed: int = 0, device=None,
                         max_pairs: int = None, noise_px: float = 0.5, mask_keep: float = 0.7, chunk: int = 128):
        """The recipe of `synthetic` (consistent_depth_amd/synthetic.py: one static height-field surface, a smooth camera path, exact
        reprojection flows + Gaussian noise, Bernoulli masks of the in-bounds pixels) evaluated ON THE DEVICE with torch in float64 --
        BASELINE configs[3]'s 1000-frame clip (2979 pairs, 7.2 GB) in seconds instead of the host generator's ~2 minutes per rank.
        Same cameras and surface as `synthetic(seed)` (they come from the same host RNG stream); colours, flow noise and masks come
        from a torch generator seeded with `seed`, so they are NOT the host generator's numbers (a different, equally seeded clip:
        every rank generates the same one).  Plumbing for benchmarks and tests -- synthetic data is not part of the product path."""
        from .. import synthetic as syn
        from ..utils import frame_range as fr, frame_sampling as fs
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        rng = np.random.default_rng(seed)
        K = syn.clip_intrinsics(H, W)
        extr = syn.camera_path(n_frames, rng, step=0.01, max_angle=0.15)
        surf = syn.Surface(rng)
        f64 = dict(dtype=torch.float64, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        y, x = torch.meshgrid(torch.arange(H, **f64), torch.arange(W, **f64), indexing="ij")
        ray = torch.stack([(x - K[2]) / K[0], -(y - K[3]) / K[1], -torch.ones_like(x)], 0)           # (3,H,W)
        E = torch.as_tensor(extr, **f64)                                                              # (N,3,4)
        sk, sph, samp = torch.as_tensor(surf.k, **f64), torch.as_tensor(surf.phase, **f64), torch.as_tensor(surf.amp, **f64)

        def depth_below(px, py):
            d = torch.full_like(px, surf.mid)
            for w in range(sk.shape[0]):
                d = d + samp[w] * torch.sin(sk[w, 0] * px + sk[w, 1] * py + sph[w])
            return d

        gt = torch.empty(n_frames, H, W, **f64)
        for s0 in range(0, n_frames, chunk):          # render_depth: fixed-point ray / height-field intersection, 12 iterations
            R, t = E[s0:s0 + chunk, :, :3], E[s0:s0 + chunk, :, 3]
            d = torch.einsum("nij,jhw->nihw", R, ray)
            s = torch.full((R.shape[0], H, W), surf.mid, **f64)
            for _ in range(12):
                px, py = t[:, 0, None, None] + s * d[:, 0], t[:, 1, None, None] + s * d[:, 1]
                s = (depth_below(px, py) + t[:, 2, None, None]) / (-d[:, 2])
            gt[s0:s0 + chunk] = s
        color = torch.rand(n_frames, 3, H, W, generator=gen, device=dev, dtype=torch.float32)
        pairs = sorted(fs.SamplePairs.to_one_way(fs.sample_pairs(fr.FrameRange(fr.OptionalSet(), n_frames), flow_ops)))
        if max_pairs:
            pairs = pairs[:max_pairs]
        P = len(pairs)
        flows = torch.empty(P, 2, 2, H, W, dtype=torch.float32, device=dev)
        masks = torch.empty(P, 2, 1, H, W, dtype=torch.uint8, device=dev)
        pa = torch.as_tensor([p[0] for p in pairs], device=dev)
        pb = torch.as_tensor([p[1] for p in pairs], device=dev)
        for s0 in range(0, P, chunk):
            for k, (ia, ib) in enumerate(((pa[s0:s0 + chunk], pb[s0:s0 + chunk]), (pb[s0:s0 + chunk], pa[s0:s0 + chunk]))):
                Rr, tr, Rt, tt = E[ia, :, :3], E[ia, :, 3], E[ib, :, :3], E[ib, :, 3]
                p3 = ray[None] * gt[ia][:, None]                                                  # (n,3,H,W) camera-space points
                world = torch.einsum("nij,njhw->nihw", Rr, p3) + tr[:, :, None, None]
                q = torch.einsum("nji,njhw->nihw", Rt, world - tt[:, :, None, None])
                fx = K[0] * q[:, 0] / (-q[:, 2]) + K[2] - x
                fy = -(K[1] * q[:, 1] / (-q[:, 2])) + K[3] - y
                f = torch.stack([fx, fy], 1) + noise_px * torch.randn(fx.shape[0], 2, H, W, generator=gen, **f64)
                inb = (x + f[:, 0] >= 0) & (x + f[:, 0] <= W - 1) & (y + f[:, 1] >= 0) & (y + f[:, 1] <= H - 1)
                keep = torch.rand(fx.shape[0], H, W, generator=gen, device=dev) < mask_keep
                flows[s0:s0 + chunk, k] = f.float()
                masks[s0:s0 + chunk, k, 0] = (keep & inb).to(torch.uint8)
        store = cls(color, flows, masks, np.tile(K, (n_frames, 1)), extr, [list(p) for p in pairs], device=dev)
        store.gt_depth = gt.float().cpu().numpy()
        return store

    def scale_scene_(self, factor: float):
        """Multiply the scene's metric scale by `factor`: camera translations (and the synthetic ground-truth depth, if any)
        scale, flows/masks/intrinsics do not depend on it.  This is what the reference's scale-calibration stage does to the
        COLMAP cameras so that the geometry matches the depth network's initial prediction (scale_calibration.py:305-313:
        `extrinsics[..., -1] /= mean_scale`)."""
        self.extrinsics[..., 3] *= float(factor)
        if getattr(self, "gt_depth", None) is not None:
            self.gt_depth = self.gt_depth * float(factor)
        return self

    def __len__(self):
        return self.flows.shape[0]

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.color, self.flows, self.masks))

    def pair_indices(self):
        """[[frame_i, frame_j], ...] in original frame numbering, store order."""
        pf = self.pair_frames.cpu().numpy()
        return [[self.frame_ids[a], self.frame_ids[b]] for a, b in pf]

    def _store_desc(self):
        if self._desc is None:
            F, _, H, W = self.color.shape
            d = _StoreDesc(self.color.data_ptr(), self.flows.data_ptr(), self.masks.data_ptr(), self.intrinsics.data_ptr(),
                           self.extrinsics.data_ptr(), self.pair_frames.data_ptr(), self._frame_ids_dev.data_ptr(),
                           self.mask_sums.data_ptr(), self.tile_windows.data_ptr(), self.tile_windows.shape[1], F, len(self), H, W, 1, 0)
            self._desc = d
        return self._desc

    def new_batch_buffers(self, B: int):
        """(images, metadata) tensors of a batch of B pairs, in the layout `batch` returns (contents undefined)."""
        _, _, H, W = self.color.shape
        dev = self.device
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        images = f(B, 2, 3, H, W)
        metadata = {
            "extrinsics": f(B, 2, 3, 4),
            "intrinsics": f(B, 2, 4),
            "geometry_consistency": {
                "indices": torch.empty(B, 2, dtype=torch.int64, device=dev),
                "flows": [f(B, 2, H, W), f(B, 2, H, W)],
                "masks": [f(B, 1, H, W), f(B, 1, H, W)],
                "mask_sums": f(B, 2),
                "tile_windows": torch.empty(B, self.tile_windows.shape[1], dtype=torch.uint8, device=dev),
            },
        }
        return images, metadata

    def gather_into(self, pair_ids: torch.Tensor, images: torch.Tensor, metadata: dict):
        """Fill existing batch buffers (e.g. the static inputs of a captured step graph) with the pairs `pair_ids`
        (int64 tensor on the device) -- one launch, no host synchronisation."""
        if pair_ids.dtype != torch.int64 or not pair_ids.is_cuda or not pair_ids.is_contiguous():
            raise ValueError("pair_ids must be a contiguous int64 tensor on the HIP device")
        B = pair_ids.numel()
        geom = metadata["geometry_consistency"]
        tensors = [images, geom["flows"][0], geom["flows"][1], geom["masks"][0], geom["masks"][1], metadata["intrinsics"],
                   metadata["extrinsics"], geom["indices"], geom["mask_sums"], geom["tile_windows"]]
        _, _, H, W = self.color.shape
        shapes = [(B, 2, 3, H, W), (B, 2, H, W), (B, 2, H, W), (B, 1, H, W), (B, 1, H, W), (B, 2, 4), (B, 2, 3, 4), (B, 2), (B, 2),
                  (B, self.tile_windows.shape[1])]
        for t, shp in zip(tensors, shapes):
            if tuple(t.shape) != shp or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"batch buffer of shape {tuple(t.shape)}: expected contiguous {shp} on {self.device}")
        out = _BatchDesc(*[t.data_ptr() for t in tensors])
        rc = _native.lib().cd_gather_pairs(ctypes.byref(self._store_desc()), pair_ids.data_ptr(), B, ctypes.byref(out),
                                           _native.stream_ptr(self.device))
        _native.check(rc, "cd_gather_pairs")
        return images, metadata

    def batch(self, pair_ids):
        ids = torch.as_tensor(pair_ids, dtype=torch.int64, device=self.device).contiguous()
        images, metadata = self.new_batch_buffers(ids.numel())
        return self.gather_into(ids, images, metadata)

    @staticmethod
    def load_directory(path: str, meta_file: str) -> dict:
        """The reference's on-disk layout (see loaders/video_dataset.py) as host arrays in store order: the constructor's
        arguments.  Pairs are the unique ordered pairs of flow_list.json IN THE REFERENCE'S DATASET ORDER
        (video_dataset.py:108-125: that order is the validation sweep's, and with train-mode BatchNorm in the sweep it decides
        which pairs share batch statistics and at which batch a frame is exported); colour rows are the frames those pairs use."""
        from . import video_dataset as vd
        ds = vd.VideoDataset(path, meta_file)
        pairs = [tuple(p) for p in ds.flow_indices]
        frames = sorted({f for p in pairs for f in p})
        row = {f: r for r, f in enumerate(frames)}
        color = np.stack([vd.load_color(ds.color_fmt.format(f)).numpy() for f in frames])
        flows = np.stack([np.stack([vd.load_flow(ds.flow_fmt.format(a, b)).numpy() for a, b in ((i, j), (j, i))])
                          for i, j in pairs])
        masks = np.stack([np.stack([vd.load_mask(ds.mask_fmt.format(a, b)).numpy() for a, b in ((i, j), (j, i))])
                          for i, j in pairs])
        return dict(color=color, flows=flows, masks=masks, intrinsics=ds.intrinsics.numpy()[frames],
                    extrinsics=ds.extrinsics.numpy()[frames], pair_frames=[[row[i], row[j]] for i, j in pairs], frame_ids=frames)

    @classmethod
    def from_directory(cls, path: str, meta_file: str, device=None):
        """Load the reference's on-disk layout into HBM."""
        return cls(device=device, **cls.load_directory(path, meta_file))

    @classmethod
    def synthetic(cls, n_frames: int, H: int, W: int, flow_ops=("hierarchical2",), seed: int = 0, device=None,
                  max_pairs: int = None):
        """Synthetic clip with the reference's pair sampling (BASELINE configs 3/4)."""
        from .. import synthetic as syn
        from ..utils import frame_range as fr, frame_sampling as fs
        video = syn.make_video(n_frames, H, W, seed)
        rng = np.random.default_rng(seed + 1)
        pairs = sorted(fs.SamplePairs.to_one_way(fs.sample_pairs(fr.FrameRange(fr.OptionalSet(), n_frames), flow_ops)))
        if max_pairs:
            pairs = pairs[:max_pairs]
        flows = np.empty((len(pairs), 2, 2, H, W), np.float32)
        masks = np.empty((len(pairs), 2, 1, H, W), np.float32)
        for p, (i, j) in enumerate(pairs):
            (f0, m0), (f1, m1) = syn.video_pair_data(video, i, j, rng)
            flows[p, 0], flows[p, 1], masks[p, 0], masks[p, 1] = f0, f1, m0, m1
        store = cls(video["color"], flows, masks, video["intrinsics"], video["extrinsics"],
                    [list(p) for p in pairs], device=device)
        store.gt_depth = video["gt_depth"]
        return store
