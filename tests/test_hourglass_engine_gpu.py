"""GPU: the HIP hourglass engine (forward + explicit backward) against the same network run by PyTorch
autograd on the CPU (fp64) -- pred_d, every parameter gradient, BatchNorm running statistics.
The CNN itself is "parity unpinned" w.r.t. the un-vendored upstream network (see oracle/hourglass_ref.py);
this test pins the HIP engine to the restated architecture."""
import numpy as np
import os

import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.double() - b.double()).abs().sum().item() / max(1e-30, b.double().abs().sum().item())


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (2, 32, 48), (2, 384, 224), (8, 384, 224)],
                         ids=["2x64x96", "2x32x48", "fullres_2x384x224", "baseline_8x384x224"])
def test_engine_matches_autograd(N, H, W):
    """The last two cases run at the BASELINE resolution (384x224): the XCD-aware tile mapping, the level streams, the timed
    launch shapes and the wide-1x1 / few-input-channel weight-gradient plans only exist at this size.  The fp64 CPU reference
    of the full BS4 batch of 8 images takes ~5 minutes of host time and tens of GB: it is a committed golden
    (tests/golden/engine_ref_8x384x224.npz, written by tests/bg_reference.py --golden from the seeds used here; every tensor
    sampled as flat[::stride], the engine's tensors are sampled the same way below); CD_AMD_TEST_LIVE_ENGINE_REF=1 computes it
    live instead (background process in a full session, inline on its own)."""
    import torch
    import conftest
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    from consistent_depth_amd.monodepth.hourglass_engine import HourglassEngine
    import bg_reference
    bg = conftest.background_engine_reference() if N == 8 else None
    sampled = isinstance(bg, dict) and "sampled" in bg
    pick = (lambda t: bg_reference.sample(t)) if sampled else (lambda t: t)      # the golden holds flat[::stride] of every tensor
    # The fp64 autograd reference of 8 images at 384x224 keeps ~25 GB of activations and peaks well above that (a GPU box was lost
    # in round 4 when the bar was lowered to 32 GB: a host that dies takes the whole run with it).  A host with less than 48 GB free
    # FAILS this test (it is the only whole-network gradient check at the headline shape; a silent skip on the driver's box would
    # read as "covered").  CD_AMD_ALLOW_MEM_SKIP=1 turns the failure into a skip on a small development host.
    def no_memory():
        msg = "less than 48 GB of free host memory: the fp64 reference at the BASELINE shape (8x384x224) cannot run"
        if os.environ.get("CD_AMD_ALLOW_MEM_SKIP") == "1":
            pytest.skip(msg)
        pytest.fail(msg + " (CD_AMD_ALLOW_MEM_SKIP=1 skips instead)")
    if bg == "skip":
        no_memory()
    if N * H * W > 100000 and bg is None:
        import psutil
        if psutil.virtual_memory().available < 48e9:
            no_memory()
    torch.manual_seed(0)
    ref = HourglassModel().double()
    net = HourglassModel()
    net.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    net.cuda().train()
    ref.train()
    x = torch.rand(N, 3, H, W, dtype=torch.float64)
    dpred = torch.randn(N, 1, H, W, dtype=torch.float64)
    if bg is None:
        pred_ref, _ = ref(x)
        pred_ref.backward(dpred)
        gref = {n: p.grad for n, p in ref.named_parameters()}
        stat_ref = {k: v for k, v in ref.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    else:    # same seed, same construction order: tests/bg_reference.py::inputs
        pred_ref = torch.as_tensor(bg["pred"])
        gref = {n: (torch.as_tensor(bg["grad:" + n]) if "grad:" + n in bg else None) for n, _ in ref.named_parameters()}
        stat_ref = {k[len("stat:"):]: torch.as_tensor(v) for k, v in bg.items() if k.startswith("stat:")}

    eng = HourglassEngine(net)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    pred = eng.forward(x.float().cuda())
    assert pred.requires_grad
    assert _rel(pick(pred.detach().cpu()), pred_ref.detach()) < 2e-4
    pred.backward(dpred.float().cuda())
    torch.cuda.synchronize()
    # fp32 noise floor of autograd itself on this (deep, BatchNorm-heavy) network: same net in fp32 on the CPU
    big = N * H * W > 400000
    g32 = None
    if not big:    # (at the BASELINE shape a second CPU pass costs minutes: its measured distances are the constants below)
        ref32 = HourglassModel()
        ref32.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items() if "running" not in k and "num_batches" not in k}, strict=False)
        ref32.train()
        p32, _ = ref32(x.float())
        p32.backward(dpred.float())
        g32 = dict(ref32.named_parameters())
    errs = []
    for name, p in net.named_parameters():
        g = gref[name]
        if name.startswith("uncertainty_layer") or g is None:
            continue
        is_bias_before_bn = name.endswith(".bias") and not name.startswith("pred_layer") and name != "seq.1.bias"
        if is_bias_before_bn:
            # mathematically zero; autograd produces round-off noise, the engine exact zeros
            assert p.grad.abs().max().item() == 0.0
            continue
        # torch-fp32 autograd at 8x384x224, measured once (profiles/parity_engine_r02.txt): median 1.07e-2, worst 1.25e-2
        errs.append((_rel(pick(p.grad.cpu()), g), _rel(g32[name].grad, g) if g32 is not None else 1.07e-2, name))
    errs.sort(reverse=True)
    for e, e32, name in errs[:8]:
        print(f"  {name:45s} engine {e:.2e}   torch-fp32 {e32:.2e}")
    med = sorted(e for e, _, _ in errs)[len(errs) // 2]
    med32 = sorted(e for _, e, _ in errs)[len(errs) // 2]
    print(f"  median engine {med:.2e}  median torch-fp32 {med32:.2e}  (N={N}, {H}x{W})")
    # The engine must be in the same noise class as fp32 autograd.  At random init this network is chaotic in the
    # ReLU masks (train-mode BN, ~100 layers): a forward round-off of 1e-6 flips a few masks and moves single
    # parameter gradients by up to ~1e-3, for autograd and the engine alike but not on the same parameters.  So:
    # nearly every parameter within 4x of autograd's own error, none beyond 4x / 3e-3 (a wrong tap, scale or mask
    # convention shows up as >= 1e-1), and the medians within 2x.
    within = [e < max(4 * e32, 2e-4) for e, e32, _ in errs]
    assert sum(within) >= 0.95 * len(within), [(n, e, e32) for (e, e32, n), ok in zip(errs, within) if not ok][:8]
    for e, e32, name in errs:
        assert e < max(4 * e32, 3e-3), (name, e, e32)
    assert med < 2 * med32 + 1e-4, (med, med32)
    # BatchNorm running statistics follow nn.BatchNorm2d
    sd = net.state_dict()
    assert len(stat_ref) == 2 * 155
    for k, v in stat_ref.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    from gpu_util import report
    report(f"engine_vs_fp64_autograd[{N}x{H}x{W}]", pred_rel_l1=_rel(pick(pred.detach().cpu()), pred_ref.detach()), grad_median=med,
           grad_worst=errs[0][0], torch_fp32_grad_median=med32, reference="golden (sampled)" if sampled else ("background" if bg is not None else "inline"))


def test_engine_under_both_conv_arithmetics():
    """cd_set_conv_arith: the same network, input and upstream gradient through the split-bf16 kernels (default) and through the
    fp32 matrix instruction -- two different accumulation orders of the same fp32-accurate convolution, so the results agree to
    the noise class of the engine-vs-autograd comparison above (and each mode is bit-reproducible on its own)."""
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    from consistent_depth_amd.monodepth.hourglass_engine import HourglassEngine
    lib = _native.lib()
    before = lib.cd_get_conv_arith()
    torch.manual_seed(2)
    net = HourglassModel().cuda().train()
    eng = HourglassEngine(net)
    x = torch.rand(2, 3, 64, 96, device="cuda")
    dpred = torch.randn(2, 1, 64, 96, device="cuda")
    out = {}
    try:
        for mode in (2, 0, 2):
            lib.cd_set_conv_arith(mode)
            for p in net.parameters():
                p.grad = torch.zeros_like(p)
            pred = eng.forward(x)
            pred.backward(dpred)
            torch.cuda.synchronize()
            res = (pred.detach().clone(), torch.cat([p.grad.flatten() for p in net.parameters()]))
            if mode in out:   # the third pass repeats the first mode: bitwise
                assert torch.equal(res[0], out[mode][0]) and torch.equal(res[1], out[mode][1])
            out[mode] = res
    finally:
        lib.cd_set_conv_arith(before)
    dp, dg = _rel(out[2][0], out[0][0]), _rel(out[2][1], out[0][1])
    print(f"  split vs fp32 arithmetic: pred {dp:.2e}  grads {dg:.2e}")
    assert dp < 2e-4 and dg < 2e-2


def test_engine_eval_mode_and_no_grad():
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    from consistent_depth_amd.monodepth.hourglass_engine import HourglassEngine
    torch.manual_seed(1)
    net = HourglassModel()
    # make the running statistics non-trivial
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    ref = HourglassModel().double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in net.state_dict().items()})
    net.cuda().eval()
    ref.eval()
    x = torch.rand(1, 3, 32, 32, dtype=torch.float64)
    eng = HourglassEngine(net)
    with torch.no_grad():
        pred = eng.forward(x.float().cuda())
    assert not pred.requires_grad
    assert _rel(pred.cpu(), ref(x)[0].detach()) < 2e-4


def test_c_handle_engine_matches_python_engine():
    """cd_hourglass_* (csrc/hourglass.hip): the plan in C++ behind one handle.  Same kernels, same buffers layout, same
    fixed summation orders as the Python orchestration -> pred, every parameter gradient and the running statistics agree
    to the last bit or nearly (launch shapes differ: results do not depend on them); parameter layout = FlatAdam's."""
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    from consistent_depth_amd.monodepth.hourglass_c import CHourglass
    from consistent_depth_amd.monodepth.hourglass_engine import HourglassEngine
    from consistent_depth_amd.optimizer import FlatAdam
    torch.manual_seed(0)
    N, H, W = 2, 64, 96
    net = HourglassModel().cuda().train()
    for m in net.modules():   # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    ce = CHourglass(N, H, W)
    ce.load_module(net)
    opt = FlatAdam(list(net.parameters()), lr=4e-4)          # re-homes the parameters: the layout the C engine mirrors
    layout = ce.param_layout()
    assert [o for o, _ in layout] == opt._offsets and ce.n_param == opt.flat_param.numel()
    x = torch.rand(N, 3, H, W, device="cuda")
    dpred = torch.randn(N, 1, H, W, device="cuda")
    eng = HourglassEngine(net)
    opt.zero_grad()
    pred_py = eng.forward(x)
    pred_py.backward(dpred)
    ce.zero_grad()
    pred_c = ce.forward(x, training=True)
    ce.backward(dpred)
    torch.cuda.synchronize()
    assert _rel(pred_c, pred_py.detach()) < 1e-6
    g_c, g_py = ce.grads(), opt.flat_grad
    assert _rel(g_c, g_py) < 1e-5, _rel(g_c, g_py)
    worst = max(_rel(g_c[o:o + int(torch.tensor(s).prod())], g_py[o:o + int(torch.tensor(s).prod())])
                for (o, s) in layout if g_py[o:o + int(torch.tensor(s).prod())].abs().sum() > 0)
    print(f"  C engine vs Python engine: pred rel {_rel(pred_c, pred_py.detach()):.2e}, grads rel {_rel(g_c, g_py):.2e}, worst tensor {worst:.2e}")
    assert worst < 1e-4
    # running statistics after one training forward, and eval mode
    _, bn_c = ce.state()
    bn_py = torch.cat([torch.cat([m.running_mean.detach().float().cpu(), m.running_var.detach().float().cpu()])
                       for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)])
    assert _rel(bn_c, bn_py) < 1e-6
    net.eval()
    with torch.no_grad():
        e_py = eng.forward(x)
    e_c = ce.forward(x, training=False)
    assert _rel(e_c, e_py) < 1e-6
    ce.close()


def _inception_of(net, kind):
    from consistent_depth_amd.monodepth import hourglass as HG
    return next(m for m in net.modules() if isinstance(m, HG.Inception) and m.kind == kind)


@pytest.mark.parametrize("kind,N,H,W", [("A2", 2, 384, 224), ("A", 4, 384, 224), ("F", 8, 96, 56), ("E", 8, 48, 28), ("B2", 8, 192, 112)],
                         ids=["A2_2x384x224", "A_4x384x224", "F_8x96x56", "E_8x48x28", "B2_8x192x112"])
def test_inception_block_matches_fp64(kind, N, H, W):
    """The gap between "one convolution at 2e-6" (test_conv_gpu.py) and "157 convolutions at the fp32-autograd noise level" (above):
    ONE inception block -- fused 1x1 entry group, three k x k branches, train-mode BatchNorm (batch statistics), ReLU -- on the
    engine's own fused launches and buffer layout, against fp64 autograd of the same block: activated output, input gradient and
    every weight gradient.  A block is two layers deep, so nothing chaotic hides a mis-scaled branch, a wrong concat offset or a
    mis-wired BatchNorm slice (each shows up as >= 1e-2).

    The loss is L = 1/2 sum_c w_c sum y_c^2 (random positive channel weights), i.e. dL/dy = w_c y on each side's OWN output -- not
    a random upstream gradient.  Measured with one (profiles/parity_blocks_random_dy_r03.txt): a weight gradient is then a sum of ~10^5..10^6
    zero-mean terms (condition number ~ sqrt(N H W)), and ONE ReLU mask that flips on a pre-activation within round-off of zero
    moves the two weight tensors of its branch by 1e-4..1e-3 -- in this engine, in its fp32-MFMA mode and in torch's fp32
    autograd alike, each on different branches.  With dL/dy = w y the gradient vanishes where the output mask flips and the sums
    carry signal, so the comparison measures the arithmetic.  Bounds: output and input gradient <= 1e-5, weight gradients <= 2e-5, or
    within 3x of torch's own fp32 autograd on the same tensor (a mid-activation mask flip still shows, in all three alike).  Shapes: the finest level (16-channel two-row tiles, k = 11), the 96x56 / 48x28 levels
    (the latency-chain launches) and 192x112."""
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel, INCEPTION
    from consistent_depth_amd.monodepth.hourglass_engine import BlockRunner, HourglassEngine
    from gpu_util import report
    torch.manual_seed(7)
    net = HourglassModel().cuda().train()
    eng = HourglassEngine(net)
    mod = _inception_of(net, kind)
    c_in = INCEPTION[kind][0]
    sd0 = {k: v.cpu() for k, v in mod.state_dict().items()}
    ref = type(mod)(kind).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd0.items()})
    ref.train()
    ref32 = type(mod)(kind)
    ref32.load_state_dict(sd0)
    ref32.train()
    g = torch.Generator().manual_seed(11)
    x_raw = torch.randn(N, c_in, H, W, generator=g, dtype=torch.float64)
    co = sum(c[-1] if len(c) > 1 else c[0] for c in INCEPTION[kind][1])
    wc = (0.5 + torch.rand(1, co, 1, 1, generator=g, dtype=torch.float64))
    a = torch.relu(x_raw).requires_grad_(True)
    y_ref = ref(a)
    (0.5 * (wc * y_ref * y_ref).sum()).backward()
    a32 = torch.relu(x_raw).float().requires_grad_(True)
    y32 = ref32(a32)
    (0.5 * (wc.float() * y32 * y32).sum()).backward()
    for p in mod.parameters():
        p.grad = torch.zeros_like(p)
    blk = BlockRunner(eng, mod, N, H, W, relu_in=True)
    y = blk.forward(x_raw.float().cuda())
    dx = blk.backward(wc.float().cuda() * y)
    torch.cuda.synchronize()
    res = {"y": _rel(y.cpu(), y_ref.detach()), "y_torch32": _rel(y32.detach(), y_ref.detach()),
           "dx": _rel(dx.cpu(), a.grad), "dx_torch32": _rel(a32.grad, a.grad)}
    gref, g32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    rows = []
    for name, p in mod.named_parameters():
        if name.endswith(".bias"):
            assert p.grad.abs().max().item() == 0.0      # in front of a train-mode BatchNorm: identically zero
            continue
        rows.append((_rel(p.grad.cpu(), gref[name].grad), _rel(g32[name].grad, gref[name].grad), name))
    worst = max(rows)
    res["dw_worst"], res["dw_worst_torch32"] = worst[0], worst[1]
    res["dw_median"] = sorted(r[0] for r in rows)[len(rows) // 2]
    res["dw_median_torch32"] = sorted(r[1] for r in rows)[len(rows) // 2]
    report(f"inception_block[{kind},{N}x{H}x{W}]", **res, worst_weight=worst[2])
    assert res["y"] <= 1e-5, res
    assert res["dx"] <= max(1e-5, 3 * res["dx_torch32"]), res
    for e, e32, name in rows:    # measured (profiles/parity_blocks_r03.txt): 8e-7 .. 1.2e-5, torch fp32: 9e-7 .. 1e-4
        assert e <= max(2e-5, 3 * e32), (name, e, e32)
    # running statistics of the block's BatchNorms follow nn.BatchNorm2d
    sd_ref, sd = ref.state_dict(), mod.state_dict()
    for k in sd_ref:
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), sd_ref[k].numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
