"""CPU: the plain-C oracle against golden vectors produced by the reference itself
(tests/golden/, made by oracle/gen_golden.py importing /root/reference)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_loss_cases, load_loss_case


@pytest.mark.parametrize("name", golden_loss_cases())
def test_loss_oracle_f64_matches_reference(oracle, name):
    batch, lr, lb, ref64, _ = load_loss_case(name)
    out = oracle.consistency_loss(batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"],
                                  batch["extrinsics"], lr, lb, dtype=np.float64)
    # fp64 restatement vs fp64 reference: only summation-order noise is allowed
    np.testing.assert_allclose(out["total"], ref64["total"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out["reprojection"], ref64["reprojection"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(out["disparity"], ref64["disparity"], rtol=1e-12, atol=1e-300)
    assert oracle.rel_l1(out["grad_depth"], ref64["grad_depth"]) < 1e-12
    scale = np.abs(ref64["grad_depth"]).max()
    assert np.abs(out["grad_depth"] - ref64["grad_depth"]).max() <= 1e-11 * scale


@pytest.mark.parametrize("name", golden_loss_cases())
def test_loss_oracle_f32_matches_reference_f32(oracle, name):
    batch, lr, lb, ref64, ref32 = load_loss_case(name)
    out = oracle.consistency_loss(batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"],
                                  batch["extrinsics"], lr, lb, dtype=np.float32)
    # fp32 noise floor of the reference itself is ~1e-7 (loss) / 7e-6 rel-L1 (grad): SURVEY.md section 4
    np.testing.assert_allclose(out["total"], ref64["total"], rtol=2e-5)
    np.testing.assert_allclose(out["reprojection"], ref64["reprojection"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(out["disparity"], ref64["disparity"], rtol=2e-5, atol=1e-7)
    tol = 5e-3 if name.startswith("stress") else 1e-4  # stress: taps hop across pixel borders in fp32
    assert oracle.rel_l1(out["grad_depth"], ref64["grad_depth"]) < tol
    # and the fp32 reference is as far from fp64 as we are (same noise class)
    ref_noise = oracle.rel_l1(ref32["grad_depth"], ref64["grad_depth"])
    assert oracle.rel_l1(out["grad_depth"], ref64["grad_depth"]) < max(tol, 10 * ref_noise)


def test_sample_oracle_matches_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "sample_b2_c3_20x28.npz"))
    out = oracle.sample(z["data"], z["uv"], dtype=np.float64)
    np.testing.assert_allclose(out, z["ref64"], rtol=1e-12, atol=1e-13)
    out32 = oracle.sample(z["data"], z["uv"], dtype=np.float32)
    np.testing.assert_allclose(out32, z["ref64"], rtol=0, atol=2e-5)


def test_adam_oracle_matches_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "adam_5steps.npz"))
    p = z["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for i, g in enumerate(z["grads"]):
        oracle.adam_step(p, np.ascontiguousarray(g), m, v, float(z["lr"]), i + 1)
        np.testing.assert_allclose(p, z["traj"][i], rtol=2e-6, atol=1e-8)


def test_empty_mask_and_border(oracle):
    """All-zero mask -> that direction contributes exactly 0 (clamp 1e-6 path)."""
    batch, lr, lb, ref64, _ = load_loss_case("stress_b2_32x48")
    assert batch["masks"][1][0].sum() == 0
    b2 = {k: (v.copy() if isinstance(v, np.ndarray) else [a.copy() for a in v]) for k, v in batch.items()}
    b2["masks"][0][:] = 0
    b2["masks"][1][:] = 0
    out = oracle.consistency_loss(b2["depth"], b2["flows"], b2["masks"], b2["intrinsics"], b2["extrinsics"], lr, lb)
    assert out["total"][0] == 0 and not out["grad_depth"].any()
