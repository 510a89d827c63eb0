"""Flow-consistency masks on the HIP device.

Mirrors the reference's utils/consistency.py (/root/reference/utils/consistency.py:53-67, used by
flow.py:199-228 `mask_valid_correspondences`): `consistent_flow_masks(flows, colors, flow_thresh, color_thresh)` takes
the two flows (H, W, 2) and the two colour images (H, W, C) of a pair as numpy arrays and returns the two boolean masks.
`consistent_flow_masks_batch` is the device-resident form for many pairs (NCHW tensors in, (B,1,H,W) 0/1 masks out --
what the loss consumes; see loaders/pair_store.py).  One fused kernel (cd_flow_consistency_masks) replaces the reference's
4 grid_sample calls + numpy reductions per pair; results are bit-identical to the reference's masks."""
from __future__ import annotations

import numpy as np
import torch

from .. import _native


def consistent_flow_masks_batch(flow_fwd, flow_bwd, color0, color1, flow_thresh=1.0, color_thresh=1.0):
    """flow_* (B,2,H,W), color* (B,C,H,W) fp32 on the HIP device -> (mask_fwd, mask_bwd), each (B,1,H,W) fp32 in {0,1}."""
    B, two, H, W = flow_fwd.shape
    if two != 2 or flow_bwd.shape != flow_fwd.shape or color0.shape != color1.shape or color0.shape[0] != B or \
            tuple(color0.shape[2:]) != (H, W):
        raise ValueError("expected flows (B,2,H,W) and colours (B,C,H,W) of the same pairs")
    f0, f1, c0, c1 = (t.float().contiguous() for t in (flow_fwd, flow_bwd, color0, color1))
    m0 = torch.empty(B, 1, H, W, dtype=torch.float32, device=f0.device)
    m1 = torch.empty_like(m0)
    rc = _native.lib().cd_flow_consistency_masks(
        _native.dev_ptr(f0, "flow_fwd"), _native.dev_ptr(f1, "flow_bwd"), _native.dev_ptr(c0, "color0"), _native.dev_ptr(c1, "color1"),
        c0.shape[1], float(flow_thresh), float(color_thresh), B, H, W, _native.dev_ptr(m0), _native.dev_ptr(m1), _native.stream_ptr(f0.device))
    _native.check(rc, "cd_flow_consistency_masks")
    return m0, m1


def consistent_flow_masks(flows, colors, flow_thresh, color_thresh, device=None):
    """The reference's call: flows = [fwd, bwd] (H,W,2), colors = [c0, c1] (H,W,C) numpy -> [mask_fwd, mask_bwd] bool (H,W)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).permute(2, 0, 1)[None].to(dev)  # noqa: E731
    colors = [np.asarray(c).reshape(c.shape[0], c.shape[1], -1) for c in colors]
    m0, m1 = consistent_flow_masks_batch(t(flows[0]), t(flows[1]), t(colors[0]), t(colors[1]), flow_thresh, color_thresh)
    return [m0[0, 0].cpu().numpy() > 0.5, m1[0, 0].cpu().numpy() > 0.5]
