#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the depth-based warp helpers, produced by RUNNING THE REFERENCE
(/root/reference/utils/geometry.py: warping_field, warp_image, calibrate_scale, depth_to_points; imported unmodified,
build container only).   python oracle/gen_golden_warp.py   -> tests/golden/warp_*.npz"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from consistent_depth_amd import synthetic  # noqa: E402

sys.path.insert(0, "/root/reference")
from utils import geometry as ref_geometry  # noqa: E402  (the reference's module)


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    for name, B, H, W, seed in (("warp_scene_b2_48x40", 2, 48, 40, 5), ("warp_scene_b3_33x57", 3, 33, 57, 6)):
        b = synthetic.make_scene_batch(B, H, W, seed=seed)
        rng = np.random.default_rng(seed)
        images = rng.random((B, 2, 3, H, W)).astype(np.float32)
        res = {"depth": b["depth"], "intrinsics": b["intrinsics"], "extrinsics": b["extrinsics"], "images": images}
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            uvs, warps, scales, pts = [], [], [], []
            for p in range(B):   # each pair is a 2-frame problem, frame 0 <-> frame 1
                t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)  # noqa: E731
                d, K, E, im = t(b["depth"][p])[:, None], t(b["intrinsics"][p]), t(b["extrinsics"][p]), t(images[p])
                uvs.append(ref_geometry.warping_field(E, K, d, [1, 0]).numpy())
                warps.append(ref_geometry.warp_image(im, d, E, K, [1, 0]).numpy())
                scales.append(float(ref_geometry.calibrate_scale(E, K, d)))
                pts.append(ref_geometry.depth_to_points(d, K).numpy())
            res.update({f"uv_{tag}": np.stack(uvs), f"warped_{tag}": np.stack(warps), f"scale_{tag}": np.array(scales),
                        f"points_{tag}": np.stack(pts)})
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **res)
        print(name, "scale", res["scale_f64"], "uv range", float(res["uv_f64"].min()), float(res["uv_f64"].max()))


if __name__ == "__main__":
    main()
