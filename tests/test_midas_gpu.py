"""The `midas2` plugin (BASELINE config 5 path) on the GPU: MiDaS-v2-shaped backbone whose convolutions (grouped, strided,
decoder) run on the hand-written MFMA kernels through ops.conv_layer.HipConv2d, reciprocal depth head fused into the HIP
loss, flat HIP Adam.  Layer parity is against an fp64 CPU convolution; the network test compares the HIP backend with the
fp64 CPU evaluation of the same module (forward + every parameter gradient) next to the PyTorch-ROCm backend's distance."""
import pytest

from tests.gpu_util import report

pytestmark = [pytest.mark.gpu]

LAYER_CASES = [  # (Cin, Cout, k, stride, groups, bias, N, H, W)
    (3, 64, 7, 2, 1, False, 2, 32, 48),      # stem
    (256, 256, 3, 1, 32, False, 2, 16, 24),  # ResNeXt 32 x 8d grouped 3x3
    (256, 256, 3, 2, 32, False, 2, 16, 24),  # ... the strided one that opens a stage
    (512, 512, 3, 2, 32, False, 1, 13, 7),   # odd extent, 16 channels per group
    (64, 256, 1, 1, 1, False, 2, 16, 24),
    (256, 512, 1, 2, 1, False, 2, 16, 24),   # down-sample shortcut
    (256, 128, 3, 1, 1, True, 2, 12, 20),    # decoder, with bias
    (32, 1, 1, 1, 1, True, 2, 12, 20),       # output head
]


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: "x".join(map(str, c)))
def test_hip_conv_layer_matches_fp64(case):
    import torch
    import torch.nn.functional as F
    from consistent_depth_amd.ops.conv_layer import HipConv2d
    Cin, Cout, k, s, G, bias, N, H, W = case
    torch.manual_seed(sum(case))
    layer = HipConv2d(Cin, Cout, k, s, (k - 1) // 2, groups=G, bias=bias).cuda()
    x = torch.randn(N, Cin, H, W, device="cuda", requires_grad=True)
    y = layer(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = layer.weight.detach().double().cpu().requires_grad_(True)
    bd = layer.bias.detach().double().cpu().requires_grad_(True) if bias else None
    yd = F.conv2d(xd, wd, bd, s, (k - 1) // 2, 1, G)
    assert yd.shape == y.shape
    yd.backward(dy.double().cpu())
    # what plain fp32 arithmetic (PyTorch-ROCm's own convolution) is away from fp64 on the same inputs: the yardstick
    x32 = x.detach().clone().requires_grad_(True)
    w32 = layer.weight.detach().clone().requires_grad_(True)
    y32 = F.conv2d(x32, w32, layer.bias.detach() if bias else None, s, (k - 1) // 2, 1, G)
    y32.backward(dy)
    rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max())  # noqa: E731
    got = {"y": rel(y, yd), "dx": rel(x.grad, xd.grad), "dw": rel(layer.weight.grad, wd.grad)}
    ref = {"y": rel(y32, yd), "dx": rel(x32.grad, xd.grad), "dw": rel(w32.grad, wd.grad)}
    if bias:
        got["db"] = rel(layer.bias.grad, bd.grad)
    report("midas_conv_layer", case="x".join(map(str, case)), **{k_: f"{v:.2e}" for k_, v in got.items()},
           **{"ref_" + k_: f"{v:.2e}" for k_, v in ref.items()})
    for name, v in got.items():
        assert v <= max(4 * ref.get(name, 0.0), 2e-6), (name, v, ref.get(name))


def test_hip_conv_layer_refuses_what_it_cannot_do():
    from consistent_depth_amd.ops.conv_layer import HipConv2d
    with pytest.raises(ValueError):
        HipConv2d(8, 8, 3, 1, 0)           # "valid" padding
    with pytest.raises(ValueError):
        HipConv2d(8, 8, 3, 3, 1)           # stride 3
    with pytest.raises(ValueError):
        HipConv2d(8, 8, 3, 1, 2, dilation=2)


def test_midas_network_hip_backend_matches_fp64():
    """Whole network, forward and every parameter gradient, HIP backend vs the fp64 CPU evaluation of the same module; the
    PyTorch-ROCm backend's distance to the same fp64 result is the yardstick (train-mode BatchNorm over tiny late-stage
    extents amplifies fp32 rounding, for both)."""
    import copy
    import torch
    from consistent_depth_amd.monodepth.midas_net import MidasNet
    torch.manual_seed(0)
    hip = MidasNet(backend="hip").cuda().train()
    ref64 = MidasNet(backend="torch")
    ref64.load_state_dict(hip.state_dict())
    ref64 = ref64.double().train()
    rocm = MidasNet(backend="torch")
    rocm.load_state_dict(hip.state_dict())
    rocm = rocm.cuda().train()
    x = torch.rand(2, 3, 64, 96)
    outs, grads = {}, {}
    for name, net, inp in (("hip", hip, x.cuda()), ("rocm", rocm, x.cuda()), ("fp64", ref64, x.double())):
        y = net(inp)
        g = torch.cos(torch.arange(y.numel(), dtype=torch.float64).reshape(y.shape)).to(y)
        (y * g).sum().backward()
        outs[name] = y.detach().double().cpu()
        # (refinenet4.resConfUnit1 has no gradient: the deepest fusion block has a single input, as upstream)
        grads[name] = {k: p.grad.detach().double().cpu() for k, p in net.named_parameters() if p.grad is not None}
    del copy

    def dist(name):
        dy = float((outs[name] - outs["fp64"]).abs().max() / outs["fp64"].abs().max())
        num = sum(float(((grads[name][k] - grads["fp64"][k]) ** 2).sum()) for k in grads["fp64"])
        den = sum(float((grads["fp64"][k] ** 2).sum()) for k in grads["fp64"])
        return dy, (num / den) ** 0.5
    assert set(grads["hip"]) == set(grads["fp64"]) and len(grads["hip"]) > 300
    hy, hg = dist("hip")
    ry, rg = dist("rocm")
    report("midas_network", hip_y=f"{hy:.2e}", hip_grad=f"{hg:.2e}", rocm_y=f"{ry:.2e}", rocm_grad=f"{rg:.2e}")
    assert hy <= max(4 * ry, 1e-5) and hg <= max(4 * rg, 1e-4)


def test_midas_one_finetune_step():
    import argparse
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    cls = get_depth_model("midas2")
    assert (cls.align, cls.learning_rate, cls.lambda_view_baseline) == (32, 0.0001, 0.0001)
    model = cls()
    assert model.backend == "hip"
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


def test_pack_pool_follows_the_weights():
    """ops/conv_layer.py::PackPool: ONE pack launch per forward covers every filter of the network (forward and transposed
    layouts).  The packed copies must track the parameters through in-place updates (what the optimiser does) and through a
    re-homing of the parameters (FlatAdam moves them into its flat buffer): forward and gradients equal the MIOpen backend's on
    the same weights, after each."""
    import torch
    from consistent_depth_amd.monodepth.midas_net import MidasNet
    torch.manual_seed(3)
    hip = MidasNet(backend="hip").cuda().train()
    rocm = MidasNet(backend="torch").cuda().train()
    x = torch.rand(2, 3, 64, 64, device="cuda")
    assert hip._pack_pool is not None and len(hip._pack_pool.layers) > 100

    def check(tag):
        rocm.load_state_dict(hip.state_dict())
        outs = []
        for net in (hip, rocm):
            for p in net.parameters():
                p.grad = None
            y = net(x)
            (y * y).sum().backward()
            outs.append((y.detach(), {k: p.grad.detach() for k, p in net.named_parameters() if p.grad is not None}))
        (ya, ga), (yb, gb) = outs
        dy = float((ya - yb).abs().max() / yb.abs().max())
        num = sum(float(((ga[k] - gb[k]) ** 2).sum()) for k in gb)
        den = sum(float((gb[k] ** 2).sum()) for k in gb)
        report("midas_pack_pool", state=tag, y=f"{dy:.2e}", grad=f"{(num / den) ** 0.5:.2e}")
        # fp32 noise of this 100-layer train-mode-BN net between two correct back ends: ~3e-4 in y, ~1e-1 in the gradients (the
        # chaos test_midas_network_hip_backend_matches_fp64 documents); filters that are 5 % stale move y by >= 1e-1
        assert dy < 2e-3 and (num / den) ** 0.5 < 0.5, (tag, dy, (num / den) ** 0.5)
    check("initial")
    with torch.no_grad():      # an optimiser step: in place, same storage
        for p in hip.parameters():
            p.mul_(1.0 + 0.05 * torch.randn_like(p))
    check("after in-place update")
    with torch.no_grad():      # FlatAdam: the parameters move to new storage
        for p in hip.parameters():
            p.data = p.data.clone()
    check("after re-homing")


def test_pooled_layer_outside_the_network_forward_sees_updated_weights():
    """A pooled HipConv2d called on its own (a sub-module call, a feature extractor) after the optimiser moved the weights: the
    fine-tuning step tells the model (`weights_updated()` -> PackPool.invalidate()), and the layer re-packs the pool instead of
    returning the buffers packed at the start of the last network forward (round 3: `fresh` was never cleared)."""
    import argparse
    import torch
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    model = get_depth_model("midas2")(seed=0)
    model.train()
    net = model.model
    pool = net._pack_pool
    layer = net.scratch.layer1_rn                         # a pooled 3x3 256 -> 256
    x = torch.rand(1, 256, 32, 32, device="cuda")
    with torch.no_grad():
        net(torch.rand(1, 3, 64, 64, device="cuda"))      # packs the pool
        assert pool.fresh
        y0 = layer(x)
        layer.weight.mul_(2.0)                            # what an optimiser step does: in place, through the same storage
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=1e-4, lambda_parameter=0, learning_rate=1e-4, optimizer="Adam")
    step = FineTuneStep(model, params, world=1)
    step._weights_updated()
    assert not pool.fresh
    with torch.no_grad():
        y1 = layer(x)                                     # re-packs (the pool is stale), then runs
    assert pool.fresh
    assert float((y1 - 2.0 * y0).abs().max()) <= 1e-5 * float(y0.abs().max())


_BLOCKS = {
    "layer1.bottleneck0 (stride 1, 64 -> 256, down-sample 1x1) @96x96": (lambda n: n.pretrained.layer1[4][0], (2, 64, 96, 96)),
    "layer2.bottleneck0 (stride 2, 256 -> 512) @96x96": (lambda n: n.pretrained.layer2[0], (2, 256, 96, 96)),
    "layer3.bottleneck5 (1024 -> 1024) @24x24": (lambda n: n.pretrained.layer3[5], (4, 1024, 24, 24)),
    "layer4.bottleneck0 (stride 2, 1024 -> 2048) @24x24": (lambda n: n.pretrained.layer4[0], (4, 1024, 24, 24)),
    "refinenet3 (two inputs, 256 ch) @24x24": (lambda n: n.scratch.refinenet3, (2, 256, 24, 24)),
    "layer3_rn (3x3 1024 -> 256) @24x24": (lambda n: n.scratch.layer3_rn, (2, 1024, 24, 24)),
}


@pytest.mark.parametrize("name", list(_BLOCKS))
def test_midas_blocks_match_fp64(name):
    """Between one convolution and the 100-layer network: single blocks of the backbone on the HIP back end against the SAME block
    of the plain-PyTorch twin evaluated in fp64 on the CPU, at the extents they have in BASELINE configs[4] (96x96 ... 12x12 for a
    384x384 input), where a train-mode BatchNorm has hundreds of samples per channel and round-off is not amplified: output, input
    gradient and EVERY parameter gradient to <= 2e-4.  A branch scaled wrongly, a stride-2 path sampled at the wrong phase, a group
    mapped to the wrong channels or a stale packed filter is an O(1) error here -- the whole-network test above cannot see such things
    through its 1e-1 noise floor."""
    import torch
    from consistent_depth_amd.monodepth.midas_net import MidasNet
    torch.manual_seed(1)
    net = MidasNet(backend="hip")
    twin = MidasNet(backend="torch")
    twin.load_state_dict(net.state_dict())
    pick, shape = _BLOCKS[name]
    net = net.cuda().train()
    block, ref = pick(net), pick(twin).double().train()
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(shape, generator=g, dtype=torch.float64) for _ in range(2 if name.startswith("refinenet") else 1)]
    res = {}
    for tag, mod, conv in (("hip", block, lambda t: t.float().cuda()), ("fp64", ref, lambda t: t.clone())):
        ins = [conv(t).requires_grad_(True) for t in xs]
        if tag == "hip":
            net._pack_pool.invalidate()          # a pooled layer called outside the network's forward re-packs the pool
        y = mod(*ins)
        w = torch.cos(torch.arange(y.numel(), dtype=torch.float64).reshape(y.shape) * 0.37).to(y)
        (0.5 * w * y * y).sum().backward()
        res[tag] = (y.detach().double().cpu(), [t.grad.detach().double().cpu() for t in ins],
                    {k: p.grad.detach().double().cpu() for k, p in mod.named_parameters() if p.grad is not None})
    rel = lambda a, b: float((a - b).abs().sum() / max(float(b.abs().sum()), 1e-300))  # noqa: E731
    dy = rel(res["hip"][0], res["fp64"][0])
    dx = max(rel(a, b) for a, b in zip(res["hip"][1], res["fp64"][1]))
    assert set(res["hip"][2]) == set(res["fp64"][2]) and res["hip"][2]
    dw = {k: rel(res["hip"][2][k], v) for k, v in res["fp64"][2].items()}
    worst = max(dw, key=dw.get)
    report(f"midas_block[{name}]", y=dy, dx=dx, dW_worst=dw[worst], worst=worst)
    assert dy < 2e-5 and dx < 2e-4 and dw[worst] < 2e-4, (dy, dx, worst, dw[worst])


@pytest.mark.parametrize("align_corners", [True, False], ids=["align_corners", "half_pixel"])
@pytest.mark.parametrize("shape", [(2, 5, 12, 12), (1, 3, 7, 9), (2, 4, 1, 6)], ids=lambda s: "x".join(map(str, s)))
def test_bilinear_up2_matches_interpolate(shape, align_corners):
    """ops.layers.bilinear_up2 -- the five x2 up-samplings of the MiDaS decoder on the hand-written gather kernels (align_corners=True in
    the fusion blocks, half-pixel centres in the output head) -- against F.interpolate in fp64: the output, and the input gradient
    (the adjoint written as a gather per input pixel: no atomics) for a random output gradient.  Odd and one-row extents included."""
    import torch
    import torch.nn.functional as F
    from consistent_depth_amd.ops.layers import bilinear_up2
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    xh = x.float().cuda().requires_grad_(True)
    y = bilinear_up2(xh, align_corners)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy.float().cuda())
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=align_corners)
    yr.backward(dy)
    assert y.shape == yr.shape
    ey = float((y.detach().double().cpu() - yr.detach()).abs().max())
    ex = float((xh.grad.double().cpu() - xr.grad).abs().max())
    report("bilinear_up2", shape="x".join(map(str, shape)), align_corners=align_corners, y=f"{ey:.2e}", dx=f"{ex:.2e}")
    # fp32 weights and products on values of a few units (the source index of align_corners=True is a rounded fp32 product): a few 1e-6
    assert ey < 6e-6 and ex < 1e-5, (ey, ex)


@pytest.mark.parametrize("relu,with_res,affine", [(True, False, True), (False, False, True), (True, True, True), (True, False, False)],
                         ids=["bn_relu", "bn", "bn_add_relu", "bn_relu_not_affine"])
@pytest.mark.parametrize("shape", [(4, 6, 12, 12), (2, 5, 7, 9)], ids=lambda s: "x".join(map(str, s)))
def test_bn_block_matches_batchnorm_fp64(shape, relu, with_res, affine):
    """ops.blocks.bn_act -- act(BatchNorm2d_train(x) [+ res]) as ONE hand-written block (csrc/bn_block.hip) -- against nn.BatchNorm2d (+ add)
    (+ ReLU) in fp64: output, input gradient, the residual's gradient, d gamma / d beta, and the running statistics + batch counter after
    the step.  H*W a multiple of 4 (16-byte path) and not."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from consistent_depth_amd.ops import blocks as B
    g = torch.Generator().manual_seed(sum(shape) + 7 * relu + 3 * with_res)
    N, C, H, W = shape
    x = torch.randn(shape, generator=g, dtype=torch.float64) * 1.7 + 0.4
    res = torch.randn(shape, generator=g, dtype=torch.float64) if with_res else None
    bn64 = nn.BatchNorm2d(C, affine=affine).double().train()
    if affine:
        with torch.no_grad():
            bn64.weight.copy_(torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
            bn64.bias.copy_(torch.randn(C, generator=g, dtype=torch.float64) * 0.3)
    bn32 = nn.BatchNorm2d(C, affine=affine).train()
    bn32.load_state_dict(bn64.state_dict())
    bn32 = bn32.cuda()
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    yr = bn64(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    dy = torch.randn(shape, generator=g, dtype=torch.float64)
    yr.backward(dy)
    xh = x.float().cuda().requires_grad_(True)
    rh = res.float().cuda().requires_grad_(True) if with_res else None
    y = B.bn_act(xh, bn32, relu, rh)
    y.backward(dy.float().cuda())
    rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).abs().max() / max(float(b.detach().abs().max()), 1e-30))  # noqa: E731
    got = {"y": rel(y, yr), "dx": rel(xh.grad, xr.grad)}
    if with_res:
        got["dres"] = rel(rh.grad, rr.grad)
    if affine:
        got["dgamma"], got["dbeta"] = rel(bn32.weight.grad, bn64.weight.grad), rel(bn32.bias.grad, bn64.bias.grad)
    got["running_mean"], got["running_var"] = rel(bn32.running_mean, bn64.running_mean), rel(bn32.running_var, bn64.running_var)
    report("bn_block", shape="x".join(map(str, shape)), relu=relu, res=with_res, affine=affine, **{k: f"{v:.2e}" for k, v in got.items()})
    assert int(bn32.num_batches_tracked) == int(bn64.num_batches_tracked) == 1
    for k, v in got.items():
        assert v < 2e-5, (k, v)      # (a ReLU mask flips where |pre-activation| < 1e-7: none in these cases)


def test_eltwise_and_maxpool_match_aten():
    """ops.blocks.relu / add / maxpool3s2 (the decoder's element-wise pieces and the stem's nn.MaxPool2d(3, 2, 1)) against the ATen ops,
    forward and backward -- BIT-identical: they move or select values, nothing is rounded.  Odd extents, -inf and ties for the pool."""
    import torch
    import torch.nn.functional as F
    from consistent_depth_amd.ops import blocks as B
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(2, 3, 9, 7, device="cuda", generator=g)
    b = torch.randn(2, 3, 9, 7, device="cuda", generator=g)
    dy = torch.randn(2, 3, 9, 7, device="cuda", generator=g)
    for fn, ref in ((lambda u, v: B.add(B.relu(u), v), lambda u, v: F.relu(u) + v),):
        u1, v1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        u2, v2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y1, y2 = fn(u1, v1), ref(u2, v2)
        y1.backward(dy); y2.backward(dy)
        assert torch.equal(y1, y2) and torch.equal(u1.grad, u2.grad) and torch.equal(v1.grad, v2.grad)
    for shape in ((2, 3, 12, 12), (1, 2, 7, 9), (1, 1, 1, 5)):
        x = torch.randn(shape, device="cuda", generator=g)
        x[..., 0, 0] = float("-inf")
        x[..., -1, :] = torch.round(x[..., -1, :])        # ties in the last row
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y1, y2 = B.maxpool3s2(x1), F.max_pool2d(x2, 3, 2, 1)
        assert y1.shape == y2.shape and torch.equal(y1, y2)
        d = torch.randn(y2.shape, device="cuda", generator=g)
        y1.backward(d); y2.backward(d)
        assert torch.equal(x1.grad, x2.grad), shape


