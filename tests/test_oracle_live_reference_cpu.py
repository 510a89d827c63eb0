"""CPU, build container only: the oracles against the reference RUN LIVE on inputs that are not in tests/golden/
(other seeds, other sizes).  Skipped wherever /root/reference does not exist (the GPU box) -- the committed goldens
cover that case.  Read-only use of the reference: its modules are imported, nothing is copied."""
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "loss")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    import torch
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))      # imported at module scope by the reference, never called here
    added = REF not in sys.path
    if added:
        sys.path.insert(0, REF)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.") or k == "loss" or k.startswith("loss.")}
    try:
        from loss.consistency_loss import ConsistencyLoss
        from utils import consistency, geometry
        yield types.SimpleNamespace(ConsistencyLoss=ConsistencyLoss, geometry=geometry, consistency=consistency, torch=torch)
    finally:
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.") or k == "loss" or k.startswith("loss.")]:
            del sys.modules[k]
        sys.modules.update(saved)
        if added:
            sys.path.remove(REF)


@pytest.mark.parametrize("B,H,W,seed,lr,lb", [(2, 21, 35, 101, 1.0, 0.1), (1, 40, 24, 102, 0.7, 0.3), (3, 16, 16, 103, 1.0, 0.0)])
def test_loss_oracle_vs_live_reference(ref, oracle, B, H, W, seed, lr, lb):
    from consistent_depth_amd import synthetic
    torch = ref.torch
    b = synthetic.make_pair_batch(B, H, W, seed=seed)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    depth = t(b["depth"]).requires_grad_(True)
    meta = {"extrinsics": t(b["extrinsics"]), "intrinsics": t(b["intrinsics"]),
            "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
    opt = types.SimpleNamespace(lambda_reprojection=lr, lambda_view_baseline=lb, lambda_parameter=0)
    total, parts = ref.ConsistencyLoss(opt)(depth, meta)
    total.backward()
    got = oracle.consistency_loss(b["depth"], b["flows"], b["masks"], b["intrinsics"], b["extrinsics"], lr, lb, dtype=np.float64)
    np.testing.assert_allclose(got["total"][0], total.item(), rtol=1e-12)
    np.testing.assert_allclose(got["reprojection"], parts["reprojection"].detach().numpy().reshape(-1), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(got["disparity"], parts["disparity"].detach().numpy().reshape(-1), rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(got["grad_depth"], depth.grad.numpy(), rtol=1e-9, atol=1e-13)


def test_mask_oracle_vs_live_reference(ref):
    from oracle import masks_oracle
    from oracle.gen_golden_masks_inputs import make_case
    for H, W, seed, wild, ft, ct in ((19, 31, 201, False, 1.0, 1.0), (28, 28, 202, True, 0.75, 0.4), (50, 37, 203, False, 2.0, 0.2)):
        flows, colors = make_case(H, W, seed, wild)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = ref.consistency.consistent_flow_masks(flows, colors, ft, ct)
        got, _ = masks_oracle.consistent_flow_masks(flows, colors, ft, ct)
        for k in range(2):
            np.testing.assert_array_equal(got[k], want[k])


def test_warp_oracle_vs_live_reference(ref):
    from consistent_depth_amd import synthetic
    from oracle import geometry_oracle as go
    torch = ref.torch
    b = synthetic.make_scene_batch(2, 27, 44, seed=301)
    img = np.random.default_rng(302).random((2, 2, 3, 27, 44))
    import warnings
    for p in range(2):
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
        d, K, E, im = t(b["depth"][p])[:, None], t(b["intrinsics"][p]), t(b["extrinsics"][p]), t(img[p])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want_w = ref.geometry.warp_image(im, d, E, K, [1, 0]).numpy()
        want_uv = ref.geometry.warping_field(E, K, d, [1, 0]).numpy()
        want_s = float(ref.geometry.calibrate_scale(E, K, d))
        w, uv = go.warp_image(img[p], b["depth"][p][:, None], b["extrinsics"][p], b["intrinsics"][p], [1, 0])
        np.testing.assert_allclose(uv, want_uv, rtol=0, atol=1e-11)
        np.testing.assert_allclose(w, want_w, rtol=0, atol=1e-12)
        assert go.calibrate_scale(b["extrinsics"][p], b["intrinsics"][p], b["depth"][p][:, None]) == pytest.approx(want_s, rel=1e-11)
