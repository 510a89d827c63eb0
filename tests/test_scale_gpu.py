"""GPU: `cd_frame_median_scales` (csrc/scale.hip: exact radix selection) bit for bit against np.median, and the product's scale stage
(consistent_depth_amd/scale_calibration.py) on disk against the golden the reference's own lines wrote
(tests/golden/scale_stage_6f_48x40.npz) -- scales.csv, metadata_scaled.npz, the scaled depth maps -- plus the round trip of its
metadata_scaled.npz through the fine-tuning path's loader."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _np_median(a, c):
    ix = np.isfinite(c)
    with np.errstate(all="ignore"):
        r = (a / c)[ix]
        return (np.median(r) if r.size else np.float32("nan")), int(ix.sum())


def test_median_kernel_is_bit_exact(torch_cuda):
    torch = torch_cuda
    from consistent_depth_amd import scale_calibration as SC
    rng = np.random.default_rng(0)
    cases = []
    for H, W, keep in ((384, 224, 0.8), (384, 224, 0.31), (37, 53, 0.9), (16, 16, 1.0), (8, 4, 0.5)):
        a = rng.uniform(0.05, 3.0, (H, W)).astype(np.float32)
        c = rng.uniform(0.05, 3.0, (H, W)).astype(np.float32)
        c[rng.random((H, W)) > keep] = np.nan
        cases.append((a, c))
    H, W = 64, 48
    base_a, base_c = rng.uniform(0.1, 2, (H, W)).astype(np.float32), rng.uniform(0.1, 2, (H, W)).astype(np.float32)
    dup = (np.round(base_a * 4) / 4).astype(np.float32), np.ones((H, W), np.float32)              # many equal ratios around the middle
    neg = (base_a - 1.0).astype(np.float32), base_c                                                # negative and positive ratios, -0.0
    infs = base_a.copy(), base_c.copy(); infs[1][:3, :7] = 0.0; infs[1][5, 5] = np.inf             # inf ratios; an infinite COLMAP value is not finite
    nan_ratio = base_a.copy(), base_c.copy(); nan_ratio[0][2, 2] = 0.0; nan_ratio[1][2, 2] = 0.0   # 0 / 0 among the selected ratios -> NaN
    empty = base_a, np.full((H, W), np.nan, np.float32)
    one = base_a, np.full((H, W), np.nan, np.float32); one[1][7, 9] = 0.5
    two = base_a, np.full((H, W), np.nan, np.float32); two[1][7, 9] = 0.5; two[1][8, 1] = 0.25
    same = cases + [dup, neg, infs, nan_ratio, empty, one, two]
    for a, c in same:
        for parity in (0, 1):
            c2 = c.copy()
            if parity and np.isfinite(c2).sum() > 2:          # flip the parity of the count: both branches of the median
                c2.flat[int(np.flatnonzero(np.isfinite(c2.ravel()))[0])] = np.nan
            want, n = _np_median(a, c2)
            s, nv, scaled = SC.frame_median_scales(torch.as_tensor(a[None]).cuda(), torch.as_tensor(c2[None]).cuda())
            got = s.cpu().numpy()[0]
            assert int(nv.item()) == n
            assert np.array_equal(np.float32(got), np.float32(want), equal_nan=True), (a.shape, n, got, want)
            with np.errstate(all="ignore"):
                assert np.array_equal(scaled.cpu().numpy()[0], a / np.float32(want), equal_nan=True)
    # a batch of frames in one launch
    A = np.stack([x[0] for x in (dup, neg, infs)]); C = np.stack([x[1] for x in (dup, neg, infs)])
    s, nv, _ = SC.frame_median_scales(torch.as_tensor(A).cuda(), torch.as_tensor(C).cuda(), want_scaled=False)
    for k in range(3):
        want, n = _np_median(A[k], C[k])
        assert np.float32(s[k].item()) == np.float32(want) and int(nv[k]) == n


def test_scale_stage_on_disk_matches_the_reference_golden(torch_cuda, tmp_path):
    from consistent_depth_amd import scale_calibration as SC
    from consistent_depth_amd.utils import image_io
    from oracle import scale_oracle as S
    z = np.load(os.path.join(GOLDEN, "scale_stage_6f_48x40.npz"))
    inv_src, inv_cmp, intr, extr = S.make_case(int(z["seed"]))
    path, out = str(tmp_path / "clip"), str(tmp_path / "out")
    S.write_case(path, out, inv_src, inv_cmp, intr, extr)
    valid = SC.calibrate_scale(path, out, sorted(inv_src))
    assert valid == set(z["scaled_frames"].tolist()) == {0, 1, 2, 4}
    assert np.array_equal(np.loadtxt(os.path.join(out, "scales.csv"), delimiter=","), z["scales_csv"])          # bit for bit
    with np.load(os.path.join(out, "metadata_scaled.npz")) as m:
        assert sorted(m.files) == ["extrinsics", "intrinsics", "scales"]
        for k in m.files:
            assert np.array_equal(m[k], z[k]), k
    for i, want in zip(z["scaled_frames"].tolist(), z["scaled"]):
        got = image_io.load_raw_float32_image(os.path.join(out, "depth_scaled_by_colmap_dense", "depth", f"frame_{i:06d}.raw"))
        assert np.array_equal(got, want, equal_nan=True), i
    assert not os.path.exists(os.path.join(out, "depth_scaled_by_colmap_dense", "depth", "frame_000003.raw"))
    # a second call finds its files (the reference's "Existing scales file loaded." / "Scaled metadata file exists." branches)
    assert SC.calibrate_scale(path, out, sorted(inv_src)) == valid


def test_metadata_scaled_round_trips_through_the_pair_store(torch_cuda, tmp_path):
    """The file the stage writes is the file the fine-tuning path reads: a synthetic clip whose metadata_scaled.npz is REPLACED by one
    written by write_scaled_metadata loads through PairStore.from_directory with the scaled translations."""
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd import scale_calibration as SC
    from consistent_depth_amd.loaders.pair_store import PairStore
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=5, H=32, W=48, seed=2)
    with np.load(os.path.join(range_dir, "metadata_scaled.npz")) as m:
        intr, extr = m["intrinsics"], m["extrinsics"]
    np.savez(os.path.join(range_dir, "metadata.npz"), intrinsics=intr, extrinsics=extr)
    os.remove(os.path.join(range_dir, "metadata_scaled.npz"))
    table = SC.write_scales_csv(os.path.join(range_dir, "scales.csv"), {0: 2.0, 1: 2.5, 3: 1.5, 4: 2.0})
    SC.write_scaled_metadata(os.path.join(range_dir, "metadata.npz"), os.path.join(range_dir, "metadata_scaled.npz"), table)
    store = PairStore.from_directory(path, os.path.join(range_dir, "metadata_scaled.npz"))
    want = extr.copy(); want[..., -1] /= 2.0
    np.testing.assert_array_equal(store.extrinsics.cpu().numpy(), want.astype(np.float32)[store.frame_ids])
    assert len(store) == len(pairs)
