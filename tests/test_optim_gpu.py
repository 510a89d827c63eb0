"""GPU parity of the flat HIP Adam and `sample` against reference-generated goldens / the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_flat_adam_matches_reference_trajectory():
    import torch
    from consistent_depth_amd import optimizer
    z = np.load(os.path.join(GOLDEN, "adam_5steps.npz"))
    p = torch.nn.Parameter(torch.tensor(z["p0"], device="cuda"))
    opt = optimizer.create("Adam", [p], float(z["lr"]), betas=(0.9, 0.999))
    for i, g in enumerate(z["grads"]):
        opt.zero_grad()
        p.grad.copy_(torch.tensor(g, device="cuda"))
        opt.step()
        np.testing.assert_allclose(p.detach().cpu().numpy(), z["traj"][i], rtol=3e-6, atol=1e-8)


def test_guarded_adam_skips_nan_and_matches_unguarded():
    import torch
    from consistent_depth_amd import optimizer
    torch.manual_seed(0)
    shapes = [(7, 3, 3, 3), (7,), (1, 130), (5,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = optimizer.create("Adam", pa, 4e-4), optimizer.create("Adam", pb, 4e-4)
    ok, nan = torch.tensor(1.0, device="cuda"), torch.tensor(float("nan"), device="cuda")
    for it in range(4):
        grads = [torch.randn(s, device="cuda") for s in shapes]
        for o, ps in ((oa, pa), (ob, pb)):
            o.zero_grad()
            for p, g in zip(ps, grads):
                p.grad.copy_(g)
        oa.step()
        if it == 2:  # a NaN-loss step in between must change nothing, not even the step counter
            before = ob.flat_param.clone()
            ob.step(guard_loss=nan)
            assert torch.equal(before, ob.flat_param)
        ob.step(guard_loss=ok)
    assert ob.step_dev.item() == 4
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-9)


def test_flat_adam_vs_torch_adam_on_hourglass_shapes():
    """Same grads -> same weights as torch.optim.Adam (what the reference instantiates)."""
    import torch
    from consistent_depth_amd import optimizer
    torch.manual_seed(1)
    shapes = [(128, 3, 7, 7), (128,), (16, 64, 11, 11), (16,), (1, 64, 3, 3), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = optimizer.create("Adam", pa, 4e-4, betas=(0.9, 0.999))
    ob = torch.optim.Adam(pb, 4e-4, betas=(0.9, 0.999))
    for _ in range(10):
        grads = [torch.randn(s, device="cuda") * 0.01 for s in shapes]
        oa.zero_grad()
        for p, q, g in zip(pa, pb, grads):
            p.grad.copy_(g)
            q.grad = g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-7)


def test_sample_matches_reference_golden(oracle):
    import torch
    from consistent_depth_amd.utils import geometry
    z = np.load(os.path.join(GOLDEN, "sample_b2_c3_20x28.npz"))
    out = geometry.sample(torch.tensor(z["data"], device="cuda"), torch.tensor(z["uv"], device="cuda"))
    np.testing.assert_allclose(out.cpu().numpy(), z["ref64"], rtol=0, atol=3e-5)
