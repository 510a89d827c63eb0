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


def test_eager_runs_are_reproducible_and_graph_replay_follows():
    """The engine reproduces itself: weight gradients are reduced in ONE fixed order (per-workgroup slices summed in split
    order, csrc/conv_wgrad.hip), the loss gradient is integer-accumulated, Adam is elementwise -- two eager runs on the same
    batches give the same losses and the same final depth (the fp64 atomics of the BatchNorm statistics are the only
    order-dependent sums left: ~1e-16 relative before the cast to fp32).  GraphedFineTuneStep (1 eager step, then capture +
    replay of the whole step) runs the same kernels on the same data, so it must land on the same trajectory."""
    batches = _batches()
    _, loss_e, depth_e = _gpu_run("hip", batches)
    _, loss_e2, depth_e2 = _gpu_run("hip", batches)
    _, loss_g, depth_g = _gpu_run("hip", batches, graph=True)
    run_to_run = _rel_l1(depth_e2, depth_e)
    print(f"\nlosses eager {loss_e}\nlosses eager again {loss_e2}\nlosses graph {loss_g}\n"
          f"depth rel-L1 eager vs eager {run_to_run:.2e}  graph vs eager {_rel_l1(depth_g, depth_e):.2e}")
    np.testing.assert_allclose(loss_e2, loss_e, rtol=1e-6)
    assert run_to_run < 1e-5
    np.testing.assert_allclose(loss_g, loss_e, rtol=1e-4)
    assert _rel_l1(depth_g, depth_e) < 1e-3
    assert loss_g[-1] < loss_g[0]


def test_run_level_parity_after_burn_in():
    """BASELINE.json's acceptance criterion -- depth maps and per-step losses within 1e-3 relative L1 of the reference's CPU
    path -- in a regime where it is a property of the ARITHMETIC and not of chaos: the comparison starts from the weights,
    BatchNorm statistics and Adam moments after a burn-in (Adam's first steps are sign-like, m/sqrt(v) = +-1: every weight
    whose gradient is at fp32 noise level moves by a full +-lr in a random direction -- the random-init start of the test
    above measures that, not the kernels).  From that state the GPU engine and the CPU restatement of the reference step
    (oracle/cpu_step.py, fp64) run the same T steps at the BASELINE image size (384x224; 2 pairs by default, the BASELINE batch of 4 with
    CD_AMD_TEST_FULL_BASELINE=1 -- see below); the reference's own fp32 run is the yardstick printed next to it."""
    import argparse
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.mannequin_challenge_model import MannequinChallengeModel
    from oracle import cpu_step, hourglass_ref
    import os
    # Default: 2 pairs = 4 images of 384x224, 2 steps -- the fp64 CPU reference is computed HERE, on the GPU box's host, and dominates the
    # test (and, at 8 images, the host's memory: torch's double convolution unfolds the whole batch).  The BASELINE batch itself (4 pairs,
    # 10 steps per epoch, 20 epochs) is covered by tests/test_loop_gpu.py::test_full_length_run_vs_fp64_and_vs_the_reference_fp32_run against an fp64
    # continuation computed offline; CD_AMD_TEST_FULL_BASELINE=1 runs this test at 4 pairs x 4 steps with the fp32 yardstick (~8 minutes).
    full = bool(os.environ.get("CD_AMD_TEST_FULL_BASELINE"))
    BURN, T, PB, PH, PW = (24, 4, 4, 384, 224) if full else (24, 2, 2, 384, 224)
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4,
                                optimizer="Adam")
    model = MannequinChallengeModel(backend="hip", seed=0)
    model.train()
    step = FineTuneStep(model, params, world=1)
    t = lambda a: torch.tensor(a, device="cuda")  # noqa: E731

    def batch(i):
        b = synthetic.make_scene_batch(PB, PH, PW, seed=500 + i)
        return np.random.default_rng(700 + i).random((PB, 2, 3, PH, PW), dtype=np.float32), b

    def meta_of(b):
        return {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
                "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}

    for i in range(BURN):
        images, b = batch(i % 6)
        step(t(images), meta_of(b))
    torch.cuda.synchronize()
    opt = step.opt
    state = {k: v.detach().cpu().clone() for k, v in model.netG.state_dict().items()}
    names = {id(p): n for n, p in model.netG.named_parameters()}
    m1, m2 = {}, {}
    for p, o in zip(opt._params, opt._offsets):
        m1[names[id(p)]] = opt.exp_avg[o:o + p.numel()].detach().cpu().clone()
        m2[names[id(p)]] = opt.exp_avg_sq[o:o + p.numel()].detach().cpu().clone()
    k0 = int(opt.step_dev.item())
    assert k0 == BURN
    run = [batch(100 + i) for i in range(T)]
    probe_images, _ = batch(999)
    losses_gpu = []
    for images, b in run:
        loss, _ = step(t(images), meta_of(b))
        losses_gpu.append(loss.item())
    with torch.no_grad():
        model.train()      # probe with batch statistics, like the reference's validation sweep
        depth_gpu = model.forward(t(probe_images)).double().cpu().numpy()

    def cpu(dtype):
        ft = cpu_step.CpuFineTuner(state, lr=4e-4, lambda_r=1.0, lambda_b=0.1, dtype=dtype)
        ft.set_adam_state(m1, m2, k0)
        import time
        import conftest
        losses = []
        for i, (images, b) in enumerate(run):
            t0 = time.monotonic()
            losses.append(float(ft.step(images, b)[0]["total"][0]))
            # the suite's time budget (tests/conftest.py): the first step calibrates this host; what is left = the other steps + the probe
            need, left = (time.monotonic() - t0) * (len(run) - 1 - i + 0.3), conftest.budget_left()
            if need > left:
                conftest.BUDGET_SKIPPED.append("test_finetune_gpu.py::test_run_level_parity_after_burn_in")
                pytest.skip(f"BUDGET-SKIP time budget: the fp64 CPU reference needs another ~{need:.0f} s on this host, {left:.0f} s are left "
                            f"(the headline-shape criterion is asserted by test_loop_gpu.py::test_full_length_run_vs_fp64_and_vs_the_reference_fp32_run "
                            f"against the committed fp64 golden; CD_AMD_TEST_BUDGET_S=0 runs this test regardless)")
        x = torch.as_tensor(probe_images, dtype=dtype).reshape(-1, 3, PH, PW)
        with torch.no_grad():
            pred, _ = hourglass_ref.forward(ft.state, x, training=True, update_running_stats=False)
        return np.array(losses), torch.exp(pred).reshape(PB, 2, PH, PW).double().numpy()

    loss64, depth64 = cpu(torch.float64)
    if full:
        loss32, depth32 = cpu(torch.float32)
    else:      # measured once at this size (profiles/parity_run_level_r02.txt): reference fp32 vs its fp64 self
        loss32, depth32 = loss64 * (1 + 5.6e-7), depth64 * (1 + 2.6e-5)
    d_gpu, d_ref = _rel_l1(depth_gpu, depth64), _rel_l1(depth32, depth64)
    l_gpu, l_ref = _rel_l1(np.array(losses_gpu), loss64), _rel_l1(loss32, loss64)
    per_step = np.abs(np.array(losses_gpu) - loss64) / np.abs(loss64)
    from gpu_util import report
    report(f"run_level[burn_in{BURN},{PB}x{PH}x{PW},{T}steps]", depth_rel_l1=d_gpu, ref_fp32_depth_rel_l1=d_ref, loss_rel_l1=l_gpu,
           ref_fp32_loss_rel_l1=l_ref, worst_step_loss_rel=float(per_step.max()))
    print(f"\nlosses gpu {losses_gpu}\nlosses cpu fp64 {loss64}\nlosses cpu fp32 {loss32}")
    assert d_gpu <= 1e-3 and l_gpu <= 1e-3 and per_step.max() <= 1e-3


def test_parameter_regulariser_reaches_the_update():
    """lambda_parameter > 0 (ParameterLoss, the L1 pull towards the initial weights): its gradient lambda * sign(p - p0) is put
    into p.grad by autograd BEFORE the engine's backward runs; the engine ACCUMULATES its weight / bias / BatchNorm-affine
    gradients, so the sum reaches Adam (round 1 overwrote it: the loss value contained the term, the update did not)."""
    import argparse
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.mannequin_challenge_model import MannequinChallengeModel
    t = lambda a: torch.tensor(a, device="cuda")  # noqa: E731
    b = synthetic.make_scene_batch(2, 64, 48, seed=5)
    images = torch.rand(2, 2, 3, 64, 48, device="cuda")
    meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
            "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
    grads, losses = {}, {}
    for lam in (0.0, 0.5):
        params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=lam, learning_rate=4e-4,
                                    optimizer="Adam")
        model = MannequinChallengeModel(backend="hip", seed=0)
        model.train()
        step = FineTuneStep(model, params, world=1)      # p0 = the initial weights
        with torch.no_grad():                            # move away from p0 deterministically, same in both runs
            gen = torch.Generator(device="cuda").manual_seed(1)
            step.opt.flat_param.add_(1e-3 * torch.randn(step.opt.flat_param.shape, device="cuda", generator=gen))
        guard, _ = step._grads(images, meta)
        grads[lam], losses[lam] = step.opt.flat_grad.clone(), guard.item()
        if lam > 0:
            sign = torch.zeros_like(step.opt.flat_param)
            for p, p0, o in zip(step.opt._params, step.criterion.parameter_loss.parameters_init, step.opt._offsets):
                sign[o:o + p.numel()] = torch.sign(p.detach() - p0).reshape(-1)
    extra = grads[0.5] - grads[0.0]
    assert losses[0.5] > losses[0.0]
    assert (extra - 0.5 * sign).abs().max().item() < 1e-5, "the regulariser's gradient did not survive the engine's backward"
    assert (extra != 0).float().mean().item() > 0.9


def test_the_step_launches_no_framework_kernels():
    """"PyTorch is plumbing, not the product": every kernel one fine-tuning step of the mc path launches (the launches a captured step graph
    replays) is this package's (namespace cd::) or a memset / copy of the HIP runtime -- no at::native kernel.  Round 5's step still had
    ~39 of them per step (22 torch.cat of the fused entry convolutions' biases, fills, a foreach-add over the BatchNorm counters, the
    loss's zeros / add / mul / sum: profiles/aten_in_step_r06.txt).  Read with torch.profiler (roctracer) over two eager steps after a
    warm-up; skipped if the profiler reports no device kernels on this stack."""
    import argparse
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.mannequin_challenge_model import MannequinChallengeModel
    from gpu_util import to_dev
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4, optimizer="Adam")
    model = MannequinChallengeModel(backend="hip", seed=0)
    model.train()
    step = FineTuneStep(model, params, world=1)
    b = synthetic.make_scene_batch(B, H, W, seed=7)
    d = to_dev(b, torch)
    images = torch.rand(B, 2, 3, H, W, device="cuda")
    meta = {"intrinsics": d["intrinsics"], "extrinsics": d["extrinsics"], "geometry_consistency": {"flows": d["flows"], "masks": d["masks"]}}
    for _ in range(3):          # plans, launch-shape timing, workspaces, tables
        step(images, meta)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    try:
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            for _ in range(2):
                step(images, meta)
            torch.cuda.synchronize()
        events = list(prof.events())
    except Exception as e:   # noqa: BLE001 -- the tracer, not the step (the step alone ran three times above)
        pytest.skip(f"torch.profiler is not usable on this stack: {type(e).__name__}: {e}")
    kernels = [e.name for e in events if str(getattr(e, "device_type", "")).endswith("CUDA") and e.name
               and not getattr(e, "is_user_annotation", False) and "#" not in e.name]       # (record_function ranges mirrored on the device timeline)
    if not any("cd::" in k for k in kernels):
        pytest.skip(f"torch.profiler reports no device kernels of this package on this stack ({len(kernels)} device events)")
    foreign = sorted({k for k in kernels if "cd::" not in k and "rocclr" not in k.lower() and not k.lower().startswith(("memcpy", "memset"))})
    assert not foreign, foreign
