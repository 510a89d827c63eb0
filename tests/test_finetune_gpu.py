"""GPU, run level: a short fine-tuning run of the product path (HIP loss + Adam, both conv back ends)
against the reference step restated on the CPU (oracle/cpu_step.py) on the SAME seeded batches.

Parity measure of BASELINE.json: relative L1 of the depth maps and of the per-step losses (budget 1e-3).
The CPU oracle is run in fp64 (ground truth) and in fp32 (what the reference actually computes).

Finding (recorded in DESIGN.md section 2): on the RANDOM-INIT network this environment allows (no pretrained
weights: no network), the 1e-3 budget is not attainable even by the reference against itself -- its fp32
run differs from its fp64 run by ~1e-1 in depth after 4 steps, because Adam's first steps are sign-like
(m/sqrt(v) = +-1), so every weight whose gradient is at fp32 noise level moves by a full +-lr in a random
direction, and the deep train-mode-BatchNorm stack amplifies that.  What CAN be asserted, and is: the GPU
path is in the same noise class as the reference's own fp32 arithmetic (depth within 3x, losses within 5x of
the reference-fp32-vs-fp64 distance), losses are finite and decrease.  The exact (non-chaotic) parts of the
step are pinned separately: loss+gradient kernel 2e-5/2e-4, Adam 3e-6, CNN forward 2e-4, CNN gradients at
the fp32-autograd noise floor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEPS, B, H, W = 4, 2, 64, 48


def _batches():
    from consistent_depth_amd import synthetic
    out = []
    for i in range(STEPS + 1):  # last one is the probe batch for the final depth comparison
        b = synthetic.make_scene_batch(B, H, W, seed=100 + i)
        images = np.random.default_rng(200 + i).random((B, 2, 3, H, W), dtype=np.float32)
        out.append((images, b))
    return out


def _cpu_run(state, batches, dtype):
    import torch
    from oracle import cpu_step
    ft = cpu_step.CpuFineTuner(state, lr=4e-4, lambda_r=1.0, lambda_b=0.1, dtype=dtype)
    losses = []
    for images, b in batches[:STEPS]:
        out, _ = ft.step(images, b)
        losses.append(float(out["total"][0]))
    # probe: train-mode forward (like the reference's validation sweep), no update
    from oracle import hourglass_ref
    x = torch.as_tensor(batches[-1][0], dtype=dtype).reshape(-1, 3, H, W)
    with torch.no_grad():
        pred, _ = hourglass_ref.forward(ft.state, x, training=True, update_running_stats=False)
    return np.array(losses), torch.exp(pred).reshape(B, 2, H, W).double().numpy()


def _gpu_run(backend, batches, graph=False):
    import argparse
    import torch
    from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
    from consistent_depth_amd.monodepth.mannequin_challenge_model import MannequinChallengeModel
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4,
                                optimizer="Adam")
    model = MannequinChallengeModel(backend=backend, seed=0)
    state = {k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}
    model.train()
    step = FineTuneStep(model, params, world=1)
    if graph:
        step = GraphedFineTuneStep(step, eager_steps=1)
    t = lambda a: torch.tensor(a, device="cuda")  # noqa: E731
    losses = []
    for images, b in batches[:STEPS]:
        meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
                "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
        loss, _ = step(t(images), meta)
        losses.append(loss.item())
    with torch.no_grad():
        depth = model.forward(t(batches[-1][0]))
    if graph:
        assert step.graphed is True, step.capture_error
    return state, np.array(losses), depth.double().cpu().numpy()


def _rel_l1(a, b):
    return float(np.abs(a - b).sum() / np.abs(b).sum())


import os

BACKENDS = ["hip"] + (["torch"] if os.environ.get("CD_AMD_TEST_TORCH_BACKEND") else [])  # torch = ~2 min of MIOpen JIT


@pytest.mark.parametrize("backend", BACKENDS)
def test_short_finetune_matches_cpu_reference(backend):
    import torch
    batches = _batches()
    state, loss_gpu, depth_gpu = _gpu_run(backend, batches)
    loss64, depth64 = _cpu_run(state, batches, torch.float64)
    loss32, depth32 = _cpu_run(state, batches, torch.float32)
    d_gpu, d_ref32 = _rel_l1(depth_gpu, depth64), _rel_l1(depth32, depth64)
    l_gpu, l_ref32 = _rel_l1(loss_gpu, loss64), _rel_l1(loss32, loss64)
    print(f"\n[{backend}] depth rel-L1 vs fp64: gpu {d_gpu:.2e}  reference-fp32 {d_ref32:.2e};  "
          f"loss rel-L1: gpu {l_gpu:.2e}  reference-fp32 {l_ref32:.2e};  losses {loss_gpu}")
    assert np.isfinite(loss_gpu).all() and loss_gpu[-1] != loss_gpu[0]
    assert loss_gpu[-1] < loss_gpu[0]
    assert d_gpu < 3 * d_ref32 + 1e-3 and l_gpu < 5 * l_ref32 + 1e-3


def test_hip_graph_replay_follows_the_eager_trajectory():
    """GraphedFineTuneStep (1 eager step, then capture + replay of the whole step: CNN forward, loss, CNN backward, Adam
    with its device-side step counter and NaN guard) must train like the eager step.  Not bit-identical: the weight
    gradients are reduced with fp32 atomics in both, and the network amplifies that (parity)."""
    batches = _batches()
    _, loss_e, depth_e = _gpu_run("hip", batches)
    _, loss_e2, depth_e2 = _gpu_run("hip", batches)
    _, loss_g, depth_g = _gpu_run("hip", batches, graph=True)
    run_to_run = _rel_l1(depth_e2, depth_e)
    print(f"\nlosses eager {loss_e}\nlosses graph {loss_g}\ndepth rel-L1 graph vs eager {_rel_l1(depth_g, depth_e):.2e}  "
          f"eager vs eager {run_to_run:.2e}")
    assert loss_g[0] == pytest.approx(loss_e[0], rel=1e-5)          # first step: same eager code
    # Step 2 sees the weights after ONE update: only the atomics' round-off separates the runs (observed 2e-7).  Later
    # steps diverge chaotically (sign-like Adam on noise-level gradients, observed up to 7e-4 in the loss and 4e-2 in the
    # depth between two EAGER runs), so they get loose bounds -- a replay bug (stale inputs, missing update) moves the
    # losses by >10 % because the batches differ (12.6, 12.3, 9.0, 6.3).
    assert loss_g[1] == pytest.approx(loss_e[1], rel=1e-3)
    np.testing.assert_allclose(loss_g[2:], loss_e[2:], rtol=5e-2)
    assert loss_g[-1] < loss_g[0]
    assert _rel_l1(depth_g, depth_e) < 3 * run_to_run + 5e-2
