"""CPU: the numpy oracle of the depth-based warp helpers (geometry.warping_field / warp_image / calibrate_scale /
depth_to_points) against goldens produced by running the reference's own functions (oracle/gen_golden_warp.py)."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "warp_*.npz")))


def test_golden_warp_cases_exist():
    assert len(GOLDEN) >= 2


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_the_reference_in_fp64(path):
    from oracle import geometry_oracle as go
    d = np.load(path)
    for p in range(d["depth"].shape[0]):
        dep, K, E = d["depth"][p][:, None], d["intrinsics"][p], d["extrinsics"][p]
        warped, uv = go.warp_image(d["images"][p], dep, E, K, [1, 0])
        np.testing.assert_allclose(uv, d["uv_f64"][p], rtol=0, atol=1e-11)
        np.testing.assert_allclose(warped, d["warped_f64"][p], rtol=0, atol=1e-12)
        np.testing.assert_allclose(go.depth_to_points(dep, K), d["points_f64"][p], rtol=0, atol=1e-12)
        assert go.calibrate_scale(E, K, dep) == pytest.approx(d["scale_f64"][p], rel=1e-11)
        # the reference's own fp32 run sits this far from its fp64 run: the floor for the fp32 HIP kernels
        assert np.abs(d["uv_f32"][p] - d["uv_f64"][p]).max() < 5e-3
