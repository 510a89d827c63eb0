"""GPU: the fused flow-consistency mask kernel (through the C ABI and the reference-shaped Python mirror) against the
golden masks produced by the reference and against the numpy oracle at BASELINE size.  Boolean output -> bit-exact."""
import numpy as np
import pytest

from test_masks_cpu import GOLDEN, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", GOLDEN, ids=[p.split("/")[-1][:-4] for p in GOLDEN])
def test_kernel_reproduces_the_reference_masks_exactly(path):
    from consistent_depth_amd.utils import consistency
    flows, colors, ft, ct, ref = load(path)
    masks = consistency.consistent_flow_masks(flows, colors, ft, ct)
    for k in range(2):
        assert masks[k].dtype == bool
        np.testing.assert_array_equal(masks[k], ref[k])


def test_batch_at_baseline_size_matches_the_oracle_exactly():
    import torch
    from consistent_depth_amd.utils import consistency
    from oracle import masks_oracle
    from oracle.gen_golden_masks_inputs import make_case
    B, H, W = 3, 384, 224
    cases = [make_case(H, W, seed=20 + b, wild=(b == 2)) for b in range(B)]
    t = lambda arrs: torch.from_numpy(np.stack([a.transpose(2, 0, 1) for a in arrs])).cuda()  # noqa: E731
    m0, m1 = consistency.consistent_flow_masks_batch(t([c[0][0] for c in cases]), t([c[0][1] for c in cases]),
                                                     t([c[1][0] for c in cases]), t([c[1][1] for c in cases]), 1.0, 0.5)
    assert m0.shape == (B, 1, H, W) and set(np.unique(m0.cpu().numpy())) <= {0.0, 1.0}
    for b, (flows, colors) in enumerate(cases):
        ref, _ = masks_oracle.consistent_flow_masks(flows, colors, 1.0, 0.5)
        np.testing.assert_array_equal(m0[b, 0].cpu().numpy() > 0.5, ref[0])
        np.testing.assert_array_equal(m1[b, 0].cpu().numpy() > 0.5, ref[1])
    # swapping the frames of the pairs swaps the masks (size-independent property)
    s1, s0 = consistency.consistent_flow_masks_batch(t([c[0][1] for c in cases]), t([c[0][0] for c in cases]),
                                                     t([c[1][1] for c in cases]), t([c[1][0] for c in cases]), 1.0, 0.5)
    assert torch.equal(s0, m0) and torch.equal(s1, m1)


def test_rejects_cpu_tensors_and_bad_shapes():
    import torch
    from consistent_depth_amd.utils import consistency
    f = torch.zeros(1, 2, 8, 8)
    c = torch.zeros(1, 3, 8, 8)
    with pytest.raises(RuntimeError):
        consistency.consistent_flow_masks_batch(f, f, c, c)
    with pytest.raises(ValueError):
        consistency.consistent_flow_masks_batch(f.cuda(), f.cuda(), c.cuda(), torch.zeros(1, 3, 8, 9).cuda())


def test_pair_store_rebuilds_its_masks_on_the_device():
    """PairStore.rebuild_masks = the reference's offline mask stage on the resident arrays: same masks as the oracle,
    and the dependent constants (loss normalisers) follow."""
    import torch
    from consistent_depth_amd.loaders.pair_store import PairStore
    from oracle import masks_oracle
    store = PairStore.synthetic(4, 32, 48, seed=3)
    before = store.masks.clone()
    store.rebuild_masks(1.0, 1.0)
    assert store.masks.shape == before.shape and set(np.unique(store.masks.cpu().numpy())) <= {0.0, 1.0}
    pf = store.pair_frames.cpu().numpy()
    for p in range(min(len(store), 3)):
        flows = [store.flows[p, k].permute(1, 2, 0).cpu().numpy() for k in range(2)]
        colors = [store.color[pf[p, k]].permute(1, 2, 0).cpu().numpy() for k in range(2)]
        ref, _ = masks_oracle.consistent_flow_masks(flows, colors, 1.0, 1.0)
        for k in range(2):
            np.testing.assert_array_equal(store.masks[p, k, 0].cpu().numpy() > 0.5, ref[k])
    torch.testing.assert_close(store.mask_sums, store.masks.float().sum((2, 3, 4)))
