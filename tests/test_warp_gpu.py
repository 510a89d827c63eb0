"""GPU: cd_warp_image / cd_depth_to_points (through the Python mirror of geometry.py) against the goldens produced by the
reference and against the fp64 oracle at BASELINE size.  Floating point: tolerances = a few times the reference's own
fp32-vs-fp64 distance on the same inputs (uv in pixels 5e-3 abs, image values 2e-3 abs, scale 1e-4 rel)."""
import numpy as np
import pytest

from test_warp_cpu import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLDEN, ids=[p.split("/")[-1][:-4] for p in GOLDEN])
def test_kernels_match_the_reference_goldens(path):
    import torch
    from consistent_depth_amd.utils import geometry
    d = np.load(path)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")  # noqa: E731
    for p in range(d["depth"].shape[0]):
        dep, K, E, im = t(d["depth"][p])[:, None], t(d["intrinsics"][p]), t(d["extrinsics"][p]), t(d["images"][p])
        uv = geometry.warping_field(E, K, dep, [1, 0]).cpu().numpy()
        warped = geometry.warp_image(im, dep, E, K, [1, 0]).cpu().numpy()
        ref32 = np.abs(d["uv_f32"][p] - d["uv_f64"][p]).max()
        assert np.abs(uv - d["uv_f64"][p]).max() < max(4 * ref32, 5e-4)
        # a tap that hops a pixel border under fp32 rounding of uv changes the value by |gradient| * eps only: continuous
        assert np.abs(warped - d["warped_f64"][p]).max() < 2e-3
        np.testing.assert_allclose(geometry.depth_to_points(dep, K).cpu().numpy(), d["points_f64"][p], rtol=2e-6, atol=1e-5)
        assert geometry.calibrate_scale(E, K, dep).item() == pytest.approx(d["scale_f64"][p], rel=1e-4)


def test_baseline_size_against_the_oracle_and_identity_warp():
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.utils import geometry
    from oracle import geometry_oracle as go
    H, W = 384, 224
    b = synthetic.make_scene_batch(1, H, W, seed=9)
    rng = np.random.default_rng(1)
    img = rng.random((2, 3, H, W)).astype(np.float32)
    dep, K, E = b["depth"][0][:, None], b["intrinsics"][0], b["extrinsics"][0]
    ref_w, ref_uv = go.warp_image(img, dep, E, K, [1, 0])
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")  # noqa: E731
    uv = geometry.warping_field(t(E), t(K), t(dep), [1, 0]).cpu().numpy()
    w = geometry.warp_image(t(img), t(dep), t(E), t(K), [1, 0]).cpu().numpy()
    assert np.abs(uv - ref_uv).max() < 5e-3 and np.abs(w - ref_w).max() < 3e-3
    # warping a frame onto itself: uv is the pixel grid; `sample` of the grid is a (W/(W-1))-scaled resampling -- equal to
    # the image only up to that documented quirk of geometry.sample, so compare with the oracle instead of the input
    uv_self = geometry.warping_field(t(E), t(K), t(dep), [0, 1]).cpu().numpy()
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    assert np.abs(uv_self[:, 0] - xs).max() < 2e-3 and np.abs(uv_self[:, 1] - ys).max() < 2e-3
    with pytest.raises(ValueError):
        geometry.warping_field(t(E), t(K), t(dep), [0, 2])
    with pytest.raises(ValueError):
        geometry.calibrate_scale(t(E)[:1], t(K)[:1], t(dep)[:1])
