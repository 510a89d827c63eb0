"""CPU: the numpy oracle of the flow-consistency mask builder against golden masks produced by the reference itself
(oracle/gen_golden_masks.py imports /root/reference/utils/consistency.py unmodified).  Bit-exact: boolean output."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "masks_*.npz")))


def load(path):
    d = np.load(path)
    return ([d["flow_fwd"], d["flow_bwd"]], [d["color0"], d["color1"]], float(d["flow_thresh"]), float(d["color_thresh"]),
            [d["mask_fwd"], d["mask_bwd"]])


def test_golden_mask_cases_exist():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_the_reference_masks_exactly(path):
    from oracle import masks_oracle
    flows, colors, ft, ct, ref = load(path)
    masks, _ = masks_oracle.consistent_flow_masks(flows, colors, ft, ct)
    for k in range(2):
        assert masks[k].dtype == bool and masks[k].shape == ref[k].shape
        np.testing.assert_array_equal(masks[k], ref[k])
    assert 0 < ref[0].mean() < 1          # the cases exercise both outcomes


def test_oracle_edge_cases():
    from oracle import masks_oracle
    H, W = 6, 9
    zero = np.zeros((H, W, 2), np.float32)
    img = np.random.default_rng(0).random((H, W, 3)).astype(np.float32)
    masks, _ = masks_oracle.consistent_flow_masks([zero, zero], [img, img], 1.0, 1.0)
    assert masks[0].all() and masks[1].all()                     # identity flow, identical images: everything is consistent
    out = zero.copy()
    out[..., 0] = W                                              # every pixel leaves the image
    masks, _ = masks_oracle.consistent_flow_masks([out, zero], [img, img], 1.0, 1.0)
    assert not masks[0].any()
