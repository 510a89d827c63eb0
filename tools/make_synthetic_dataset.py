#!/usr/bin/env python3
"""Write a synthetic clip in the reference's on-disk layout (stand-in for the ayush demo, which
needs the network): color_down/*.raw (BGR), flow/*.raw, mask/*.png, flow_list.json,
<range_dir>/metadata_scaled.npz.

    python tools/make_synthetic_dataset.py --path /tmp/syn --frames 12 --height 64 --width 48
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_dataset(path, n_frames, H, W, flow_ops=("hierarchical2",), model_type="mc", seed=0, mask_keep=0.7, noise_px=0.5,
                  step=0.01, max_angle=0.15):
    """`mask_keep` = Bernoulli keep rate of the in-bounds pixels (1.0: every in-bounds pixel is constrained), `step` / `max_angle` =
    camera motion of the clip (consistent_depth_amd/synthetic.py::make_video); the defaults are the clip every earlier golden used."""
    from consistent_depth_amd import synthetic as syn
    from consistent_depth_amd.utils import frame_range as fr, frame_sampling as fs, image_io
    video = syn.make_video(n_frames, H, W, seed, step=step, max_angle=max_angle)
    rng = np.random.default_rng(seed + 1)
    for d in ("color_down", "flow", "mask"):
        os.makedirs(os.path.join(path, d), exist_ok=True)
    for f in range(n_frames):
        bgr = video["color"][f].transpose(1, 2, 0)[..., ::-1]
        image_io.save_raw_float32_image(os.path.join(path, "color_down", f"frame_{f:06d}.raw"), bgr)
    pairs = sorted(fs.SamplePairs.to_one_way(fs.sample_pairs(fr.FrameRange(fr.OptionalSet(), n_frames), flow_ops)))
    both = []
    for i, j in pairs:
        (f0, m0), (f1, m1) = syn.video_pair_data(video, i, j, rng, noise_px=noise_px, mask_keep=mask_keep)
        for (a, b), f, m in (((i, j), f0, m0), ((j, i), f1, m1)):
            image_io.save_raw_float32_image(os.path.join(path, "flow", f"flow_{a:06d}_{b:06d}.raw"), f.transpose(1, 2, 0))
            image_io.save_mask_png(os.path.join(path, "mask", f"mask_{a:06d}_{b:06d}.png"), m[0])
            both.append([a, b])
    with open(os.path.join(path, "flow_list.json"), "w") as f:
        json.dump(both, f)
    range_dir = os.path.join(path, f"R_{'-'.join(flow_ops)}_{model_type}")
    os.makedirs(range_dir, exist_ok=True)
    np.savez(os.path.join(range_dir, "metadata_scaled.npz"), intrinsics=video["intrinsics"].astype(np.float32),
             extrinsics=video["extrinsics"].astype(np.float32), scales=np.ones((n_frames, 2), np.float32))
    return range_dir, pairs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", required=True)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=48)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rd, pairs = write_dataset(a.path, a.frames, a.height, a.width, seed=a.seed)
    print(f"wrote {a.frames} frames, {len(pairs)} pairs; range dir {rd}")
