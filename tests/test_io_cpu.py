"""CPU: on-disk contracts of the hot path's inputs/outputs (.raw codec pinned to bytes written by the
reference; dataset layout; CLI flags and defaults)."""
import os
import sys

import numpy as np

from conftest import GOLDEN, REPO


def test_raw_codec_matches_reference_bytes(tmp_path):
    from consistent_depth_amd.utils import image_io
    z = np.load(os.path.join(GOLDEN, "raw_codec.npz"))
    for key in ("hw2", "hw"):
        img, blob = z[f"img_{key}"], z[f"bytes_{key}"].tobytes()
        fn = str(tmp_path / f"{key}.raw")
        image_io.save_raw_float32_image(fn, img)
        assert open(fn, "rb").read() == blob           # we write what the reference writes
        with open(fn, "wb") as f:
            f.write(blob)
        assert np.array_equal(image_io.load_raw_float32_image(fn), img)   # and read what it wrote


def test_synthetic_dataset_layout_and_video_dataset(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd.loaders.video_dataset import VideoDataset, VideoFrameDataset
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=5, H=32, W=48, seed=1)
    assert os.path.basename(range_dir) == "R_hierarchical2_mc"
    ds = VideoDataset(path, os.path.join(range_dir, "metadata_scaled.npz"))
    assert sorted(tuple(p) for p in ds.flow_indices) == sorted(tuple(p) for p in pairs)
    images, meta = ds[0]
    assert images.shape == (2, 3, 32, 48) and images.dtype.is_floating_point
    g = meta["geometry_consistency"]
    assert g["flows"][0].shape == (2, 32, 48) and g["masks"][1].shape == (1, 32, 48)
    assert set(np.unique(g["masks"][0].numpy())) <= {0.0, 1.0}
    assert meta["intrinsics"].shape == (2, 4) and meta["extrinsics"].shape == (2, 3, 4)
    fds = VideoFrameDataset(os.path.join(path, "color_down", "frame_{:06d}.raw"), [0, 3])
    im, m = fds[1]
    assert im.shape == (3, 32, 48) and m["frame_id"] == 3


def test_cli_flags_and_defaults():
    from consistent_depth_amd.depth_fine_tuning import make_tag
    from consistent_depth_amd.params import Video3dParamsParser
    p = Video3dParamsParser().parse(["--path", "/tmp/x"])
    assert (p.size, p.align, p.batch_size, p.num_epochs, p.model_type, p.flow_ops) == (384, 16, 4, 20, "mc", ["hierarchical2"])
    assert (p.learning_rate, p.lambda_view_baseline, p.lambda_reprojection, p.lambda_parameter) == (0.0004, 0.1, 1.0, 0)
    assert make_tag(p) == "B0.1_R1.0_PL1-0_LR0.0004_BS4_Oadam"   # reference README.md:58
    q = Video3dParamsParser().parse(["--path", "/tmp/x", "--model_type", "midas2", "--frame_range", "6,6,5,8,0,2-4"])
    assert (q.align, q.learning_rate, q.lambda_view_baseline) == (32, 0.0001, 0.0001)
    assert q.frame_range.name == "0,2-6,8"


def test_registry_surface():
    import pytest
    from consistent_depth_amd.monodepth import depth_model_registry as R
    assert R.get_depth_model_list() == ["mc", "midas2", "monodepth2"]
    with pytest.raises(ValueError, match="Unsupported model type"):
        R.get_depth_model("nope")
    assert R.get_depth_model("mc").align == 16 and R.get_depth_model("monodepth2").lambda_view_baseline == 1
