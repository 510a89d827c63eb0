"""GPU smoke of the `midas2` plugin (BASELINE config 5 path): MiDaS-v2-shaped backbone through PyTorch-ROCm,
reciprocal depth head fused into the HIP loss, flat HIP Adam.  Off by default (MIOpen JIT-compiles ~100 conv
shapes, ~2 min): set CD_AMD_TEST_MIDAS=1."""
import os

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("CD_AMD_TEST_MIDAS"), reason="CD_AMD_TEST_MIDAS not set")]


def test_midas_one_finetune_step():
    import argparse
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    cls = get_depth_model("midas2")
    assert (cls.align, cls.learning_rate, cls.lambda_view_baseline) == (32, 0.0001, 0.0001)
    model = cls()
    model.train()
    assert sum(p.numel() for p in model.parameters()) > 100e6
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=1e-4, lambda_parameter=0, learning_rate=1e-4,
                                optimizer="Adam")
    step = FineTuneStep(model, params, world=1)
    b = synthetic.make_scene_batch(2, 64, 64, seed=1)
    t = lambda a: torch.tensor(a, device="cuda")  # noqa: E731
    meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
            "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
    w0 = step.opt.flat_param.clone()
    loss, parts = step(torch.rand(2, 2, 3, 64, 64, device="cuda"), meta)
    assert torch.isfinite(loss).all() and set(parts) == {"reprojection", "disparity"}
    assert not torch.equal(w0, step.opt.flat_param)
    with torch.no_grad():
        depth = model.forward(torch.rand(1, 2, 3, 64, 64, device="cuda"))
    assert depth.shape == (1, 2, 64, 64) and torch.isfinite(depth).all() and (depth > 0).all()
