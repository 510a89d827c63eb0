#!/usr/bin/env python3
"""BUILD CONTAINER ONLY.  Writes tests/golden/loop_6f_64x48.npz: the artefacts of the REFERENCE'S OWN training loop
(/root/reference/depth_fine_tuning.py `DepthFineTuner.fine_tune` + `save_depth`, run unmodified through
oracle/ref_loop.py) on the seeded 6-frame 64x48 clip of tools/make_synthetic_dataset.py, 2 epochs, BS4, from the seeded
random-init hourglass -- once in fp64 (ground truth) and once in the reference's native fp32 (the yardstick: how far the
reference is from itself).  The GPU box has no /root/reference: tests/test_loop_gpu.py regenerates the same clip and
initial weights from their seeds, replays the recorded step order and compares the product loop's artefacts with these.

    python -m oracle.gen_golden_loop
"""
import glob
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

CLIP = dict(n_frames=6, H=64, W=48, seed=3)
EPOCHS, INIT_SEED, LOOP_SEED = 2, 0, 0


def initial_state():
    """The seeded random-init hourglass (the only weights available offline), identical on every machine (CPU RNG)."""
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    torch.manual_seed(INIT_SEED)
    return HourglassModel().state_dict()


def collect(run):
    """Every numeric artefact of a finished run as flat arrays."""
    from consistent_depth_amd.utils import image_io
    out_dir = run["out_dir"]
    res = {"step_losses": np.array([l for _, _, l in run["steps"]], np.float64)}
    for fn in sorted(glob.glob(os.path.join(out_dir, "eval", "loss_*.json"))):
        with open(fn) as f:
            d = json.load(f)
        tag = os.path.basename(fn)[len("loss_"):-len(".json")]
        keys = list(d["reprojection"])
        res[f"val_{tag}_pairs"] = np.array([json.loads(k) for k in keys], np.int64)
        res[f"val_{tag}_reprojection"] = np.array([d["reprojection"][k] for k in keys], np.float64)
        res[f"val_{tag}_disparity"] = np.array([d["disparity"][k] for k in keys], np.float64)
        res[f"val_{tag}_mean"] = np.array([d["mean"]["reprojection"], d["mean"]["disparity"]], np.float64)
        frames = sorted(glob.glob(os.path.join(out_dir, "eval", f"depth_*_{tag}.raw")))
        res[f"evaldepth_{tag}"] = np.stack([image_io.load_raw_float32_image(p) for p in frames])
    frames = sorted(glob.glob(os.path.join(out_dir, "depth", "frame_*.raw")))
    if frames:
        res["depth"] = np.stack([image_io.load_raw_float32_image(p) for p in frames])
    ck = sorted(glob.glob(os.path.join(out_dir, "checkpoints", "*.pth")))[-1]
    sd = torch.load(ck, map_location="cpu")
    res["ckpt_keys"] = np.array(list(sd))
    res["ckpt_abs_sum"] = np.array([float(v.double().abs().sum()) for v in sd.values()], np.float64)
    res["ckpt_sum"] = np.array([float(v.double().sum()) for v in sd.values()], np.float64)
    return res


def main():
    import make_synthetic_dataset as msd
    from oracle import ref_loop
    tmp = tempfile.mkdtemp()
    clip = os.path.join(tmp, "clip")
    range_dir, pairs = msd.write_dataset(clip, **CLIP)
    init = initial_state()
    out = {"clip": np.array([CLIP["n_frames"], CLIP["H"], CLIP["W"], CLIP["seed"]]), "epochs": np.array(EPOCHS)}
    for name, dtype in (("ref64", torch.float64), ("ref32", torch.float32)):
        run = ref_loop.run(clip, range_dir, list(range(CLIP["n_frames"])), init, os.path.join(tmp, "work_" + name), dtype=dtype,
                           num_epochs=EPOCHS, seed=LOOP_SEED)
        if name == "ref64":
            out["order_epoch"] = np.array([e for e, _, _ in run["steps"]], np.int64)
            out["order_pairs"] = np.array([json.dumps(p) for _, p, _ in run["steps"]])
            order = [(e, p) for e, p, _ in run["steps"]]
        else:
            assert [(e, p) for e, p, _ in run["steps"]] == order, "the seeded DataLoader order must not depend on the dtype"
        for k, v in collect(run).items():
            out[f"{name}_{k}"] = v
    dst = os.path.join(REPO, "tests", "golden", "loop_6f_64x48.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
    a, b = out["ref64_step_losses"], out["ref32_step_losses"]
    print("reference fp32 vs fp64, per-step loss rel:", np.abs(a - b) / np.abs(a))


if __name__ == "__main__":
    main()
