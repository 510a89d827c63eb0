"""GPU parity of the wgrad convolution and the memory-bound layer kernels against torch on the CPU (fp64)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["split", "fp32"])
def arith(request):
    """The split-bf16 (default) and the fp32-instruction arithmetic of the convolutions (cd_set_conv_arith)."""
    from consistent_depth_amd import _native
    lib = _native.lib()
    before = lib.cd_get_conv_arith()
    assert lib.cd_set_conv_arith(2 if request.param == "split" else 0) == 0
    yield request.param
    lib.cd_set_conv_arith(before)


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [
    (2, 64, 16, 20, 36, 11), (2, 32, 32, 17, 40, 7), (2, 64, 64, 16, 32, 7), (2, 32, 64, 12, 24, 5),
    (2, 64, 32, 16, 32, 3), (2, 128, 64, 16, 40, 1), (2, 256, 32, 8, 16, 1), (1, 64, 1, 24, 40, 3), (2, 3, 128, 24, 40, 7),
    (2, 16, 16, 9, 33, 3), (2, 128, 16, 8, 32, 1), (2, 24, 40, 13, 35, 11), (3, 16, 64, 30, 66, 7), (1, 40, 24, 31, 30, 5),
])
def test_wgrad_matches_autograd(N, Cin, Cout, H, W, ks, arith):
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator().manual_seed(ks * 77 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, ks, ks, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64)
    torch.nn.functional.conv2d(torch.relu(x), w, padding=(ks - 1) // 2).backward(dy)
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
    conv.conv2d_wgrad(x.float().cuda(), dy.float().cuda(), Cin, Cout, ks, dw, ws, in_relu=True)
    ref = w.grad
    assert (dw.cpu().double() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    # accumulate mode
    conv.conv2d_wgrad(x.float().cuda(), dy.float().cuda(), Cin, Cout, ks, dw, ws, in_relu=True, accumulate=True)
    assert (dw.cpu().double() - 2 * ref).abs().max().item() < 6e-5 * ref.abs().max().item()


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(4, 1024, 1024, 24, 24), (16, 2048, 512, 12, 12), (5, 520, 1000, 20, 24), (2, 512, 2048, 48, 24)])
def test_1x1_wgrad_of_wide_filters_on_small_planes(N, Cin, Cout, H, W, arith):
    """The 1x1 weight gradient of >= 512 x 512 filters on a few thousand pixels (layer3 / layer4 of MiDaS' ResNeXt-101 encoder at
    384x384): under "split" wgrad1x1_split.hip since round 6 (csrc/wgrad_split.h::wgrad1x1_split_ok; the staged fp32 kernel before),
    under "fp32" the staged kernel -- against fp64, channel slices and the fused input transform included, two launches bit for bit."""
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin + 3, H, W, generator=g)
    dy = torch.randn(N, Cout + 5, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    act = torch.relu(x[:, 2:2 + Cin].double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = torch.einsum("nohw,nihw->oi", dy[:, 1:1 + Cout].double(), act).reshape(Cout, Cin, 1, 1)
    ws = conv.wgrad_workspace(Cout, Cin, 1, "cuda")
    got = []
    for _ in range(2):
        dw = torch.empty(Cout, Cin, 1, 1, device="cuda")
        conv.conv2d_wgrad(x.cuda(), dy.cuda(), Cin, Cout, 1, dw, ws, x_coff=2, dy_coff=1, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True)
        got.append(dw)
    assert torch.equal(got[0], got[1])
    scale = ref.abs().max().item()
    assert (got[0].cpu().double() - ref).abs().max().item() < 3e-5 * scale


@pytest.mark.parametrize("N,Cin,Cout,H,W", [
    (2, 128, 208, 130, 128),   # 14 x 8 channel tiles (13 live), odd number of 2-row tiles
    (2, 256, 160, 128, 100),   # 10 x 16, W % 4 == 0 but not a multiple of the 32-pixel tile
    (2, 128, 160, 128, 128),   # 10 x 8
    (2, 256, 112, 128, 126),   # 8 x 16 (7 live), W % 4 != 0: unpipelined staging
    (3, 120, 256, 96, 128),    # 16 x 8, Cin not a multiple of 16
    (2, 128, 128, 128, 128),   # narrow plan in both modes (4 groups only)
])
def test_wide_1x1_wgrad_matches_autograd_and_the_narrow_plan(N, Cin, Cout, H, W):
    """The wide 1x1 weight-gradient plan (all channels of dY and X of a pixel tile in one workgroup; chosen for large
    images) against autograd in fp64 and against the narrow plan (cd_debug_set_wgrad_mode bit 2)."""
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin + 3, H, W, generator=g)
    dy = torch.randn(N, Cout + 5, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.2
    act = torch.relu(x[:, 2:2 + Cin].double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = torch.einsum("nohw,nihw->oi", dy[:, 1:1 + Cout].double(), act).reshape(Cout, Cin, 1, 1)
    ws = conv.wgrad_workspace(Cout, Cin, 1, "cuda")
    out = {}
    try:
        for name, bits in (("wide", 0), ("narrow", 4)):
            _native.lib().cd_debug_set_wgrad_mode(bits)
            dw = torch.empty(Cout, Cin, 1, 1, device="cuda")
            conv.conv2d_wgrad(x.cuda(), dy.cuda(), Cin, Cout, 1, dw, ws, x_coff=2, dy_coff=1, in_scale=sc.cuda(),
                              in_shift=sh.cuda(), in_relu=True)
            out[name] = dw.cpu().double()
    finally:
        _native.lib().cd_debug_set_wgrad_mode(0)
    scale = ref.abs().max().item()
    assert (out["wide"] - ref).abs().max().item() < 3e-5 * scale
    assert (out["narrow"] - ref).abs().max().item() < 3e-5 * scale


def test_bn_normalize_and_backward_match_torch():
    import torch
    from consistent_depth_amd.ops import conv, layers
    g = torch.Generator().manual_seed(3)
    N, C, H, W, coff, ctot = 3, 24, 12, 20, 8, 40
    raw = torch.randn(N, C, H, W, generator=g, dtype=torch.float64) * 2 + 0.5
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    for affine in (False, True):
        x = raw.clone().requires_grad_(True)
        bn = torch.nn.BatchNorm2d(C, affine=affine).double()
        if affine:
            bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
        a = torch.relu(bn(x))
        dA = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
        a.backward(dA)
        buf = torch.zeros(N, ctot, H, W).cuda()
        buf[:, coff:coff + C] = raw.float().cuda()
        stats = layers.new_stats(ctot, "cuda")          # (STAT_SLOTS, ctot, 2): split the sums over three of the copies
        for slot, part in ((0, raw[:1]), (5, raw[1:2]), (layers.STAT_SLOTS - 1, raw[2:])):
            stats[slot, coff:coff + C, 0] = part.sum((0, 2, 3)).cuda()
            stats[slot, coff:coff + C, 1] = (part ** 2).sum((0, 2, 3)).cuda()
        mi = torch.zeros(ctot, 2).cuda()
        rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
        layers.bn_normalize(buf, coff, C, stats, mi, running_mean=rm, running_var=rv)
        xhat_ref = (raw - raw.mean((0, 2, 3), keepdim=True)) / torch.sqrt(raw.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
        assert (buf[:, coff:coff + C].cpu().double() - xhat_ref).abs().max().item() < 2e-5
        np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=1e-5)
        d = torch.zeros(N, ctot, H, W).cuda()
        d[:, coff:coff + C] = dA.float().cuda()
        sums = torch.zeros(2 * C, dtype=torch.float64).cuda()
        kw = {}
        if affine:
            kw = dict(gamma=gamma.float().cuda(), beta=beta.float().cuda(), dgamma=torch.zeros(C).cuda(), dbeta=torch.zeros(C).cuda())
        layers.bn_relu_bwd(d, coff, buf, coff, C, mi, sums, **kw)
        # the pass-free variant: (scale, shift) from the statistics, raw tensor + apply-on-load semantics
        raw_buf = torch.zeros(N, ctot, H, W).cuda()
        raw_buf[:, coff:coff + C] = raw.float().cuda()
        mi2, sc, sh = torch.zeros(ctot, 2).cuda(), torch.ones(ctot).cuda(), torch.zeros(ctot).cuda()
        layers.bn_finalize(stats, coff, C, N * H * W, mi2, sc, sh, gamma=kw.get("gamma"), beta=kw.get("beta"))
        act = torch.relu(raw_buf[:, coff:coff + C] * sc[coff:coff + C].view(1, -1, 1, 1) + sh[coff:coff + C].view(1, -1, 1, 1))
        assert (act.cpu().double() - a.detach()).abs().max().item() < 3e-5
        d2 = torch.zeros(N, ctot, H, W).cuda()
        d2[:, coff:coff + C] = dA.float().cuda()
        kw2 = dict(kw)
        if affine:   # CD_BN_BWD_OVERWRITE_AFFINE: unzeroed gradient buffers are assigned, not accumulated into
            kw2.update(dgamma=torch.full((C,), 123.0).cuda(), dbeta=torch.full((C,), -7.0).cuda(), overwrite_affine=True)
        layers.bn_relu_bwd(d2, coff, raw_buf, coff, C, mi2, torch.zeros(2 * C, dtype=torch.float64).cuda(), scale=sc, shift=sh, **kw2)
        assert (d2[:, coff:coff + C].cpu().double() - x.grad).abs().max().item() < 3e-5 * x.grad.abs().max().item()
        assert (d[:, coff:coff + C].cpu().double() - x.grad).abs().max().item() < 3e-5 * x.grad.abs().max().item()
        if affine:
            np.testing.assert_allclose(kw["dgamma"].cpu().numpy(), bn.weight.grad.numpy(), rtol=2e-5, atol=1e-4)
            np.testing.assert_allclose(kw["dbeta"].cpu().numpy(), bn.bias.grad.numpy(), rtol=2e-5, atol=1e-4)
            np.testing.assert_allclose(kw2["dgamma"].cpu().numpy(), bn.weight.grad.numpy(), rtol=2e-5, atol=1e-4)
            np.testing.assert_allclose(kw2["dbeta"].cpu().numpy(), bn.bias.grad.numpy(), rtol=2e-5, atol=1e-4)
            g0 = kw["dgamma"].clone()       # default: accumulate (a second call doubles the stored gradient)
            d3 = torch.zeros(N, ctot, H, W).cuda()
            d3[:, coff:coff + C] = dA.float().cuda()
            layers.bn_relu_bwd(d3, coff, buf, coff, C, mi, torch.zeros(2 * C, dtype=torch.float64).cuda(), **kw)
            np.testing.assert_allclose(kw["dgamma"].cpu().numpy(), 2 * g0.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_pool_upsample_add_and_adjoints():
    import torch
    from consistent_depth_amd.ops import layers
    g = torch.Generator().manual_seed(5)
    N, C, h, w = 2, 6, 7, 12
    x = torch.randn(N, C, 2 * h, 2 * w, generator=g, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.avg_pool2d(torch.relu(x), 2)
    dy = torch.randn(N, C, h, w, generator=g, dtype=torch.float64)
    y.backward(dy)
    out = torch.zeros(N, C + 2, h, w).cuda()
    layers.avgpool2_fwd(x.detach().float().cuda(), 0, C, out, 1, in_relu=True)
    assert (out[:, 1:C + 1].cpu().double() - y).abs().max().item() < 1e-6
    dx = torch.ones(N, C, 2 * h, 2 * w).cuda()
    layers.avgpool2_bwd(dy.float().cuda(), 0, dx, 0, C, accumulate=True)
    # gradient w.r.t. the ACTIVATED input = 0.25 * dy (the relu mask is applied by the producer's backward)
    ref = torch.nn.functional.interpolate(dy, scale_factor=2, mode="nearest") * 0.25 + 1
    assert (dx.cpu().double() - ref).abs().max().item() < 1e-6

    lo = torch.randn(N, C, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    hi = torch.randn(N, C, 2 * h, 2 * w, generator=g, dtype=torch.float64, requires_grad=True)
    o = torch.nn.functional.interpolate(torch.relu(lo), scale_factor=2, mode="bilinear", align_corners=True) + torch.relu(hi)
    do = torch.randn(N, C, 2 * h, 2 * w, generator=g, dtype=torch.float64)
    o.backward(do)
    got = torch.zeros(N, C, 2 * h, 2 * w).cuda()
    layers.upsample2x_add_fwd(lo.detach().float().cuda(), 0, C, got, 0, hi=hi.detach().float().cuda(), lo_relu=True, hi_relu=True)
    assert (got.cpu().double() - o).abs().max().item() < 2e-6
    dlo = torch.zeros(N, C, h, w).cuda()
    layers.upsample2x_bwd(do.float().cuda(), 0, dlo, 0, C, accumulate=False)
    # adjoint w.r.t. the activated low-res input = U^T do
    lo2 = torch.randn(N, C, h, w, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.interpolate(lo2, scale_factor=2, mode="bilinear", align_corners=True).backward(do)
    assert (dlo.cpu().double() - lo2.grad).abs().max().item() < 1e-5
    s = torch.zeros(C).cuda()
    layers.channel_sum(do.float().cuda(), 0, C, s)
    np.testing.assert_allclose(s.cpu().numpy(), do.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("N,C,h,w", [(2, 3, 192, 112), (1, 2, 96, 56), (2, 2, 48, 28), (1, 3, 24, 14), (1, 2, 12, 7), (2, 2, 7, 12), (1, 2, 5, 4),
                                     (1, 2, 4, 5), (1, 1, 67, 33), (1, 2, 3, 3), (1, 1, 2, 2), (1, 1, 1, 6), (1, 2, 192, 192), (1, 1, 130, 250)])
def test_streaming_layer_kernels_give_the_bits_of_the_scalar_ones(N, C, h, w):
    """Round 6: 16-byte forms of AvgPool2d(2) / UpsamplingBilinear2d(2)+add and their adjoints, an LDS-band form of the bilinear adjoint
    (csrc/layers.hip::upsample2x_bwd_band_kernel) and equal-trip-count grids for the fan-in add.  Every one must give the bits of the
    scalar kernel it replaces (cd_debug_set_layers_mode(1)): the hourglass step's bit-reproducibility and the goldens of
    tests/test_loop_gpu.py rest on it.  Sizes: the four pyramid levels of 384x224, odd / tiny / degenerate planes (the band kernel's
    fallbacks), planes wider than its LDS budget; channel offsets into wider buffers; with and without accumulate / skip tensor /
    producer affine."""
    import torch
    from consistent_depth_amd.ops import layers
    g = torch.Generator().manual_seed(100 * h + w)
    H, W = 2 * h, 2 * w
    lo = torch.randn(N, C + 2, h, w, generator=g).cuda()
    hi = torch.randn(N, C + 1, H, W, generator=g).cuda()
    do = torch.randn(N, C + 3, H, W, generator=g).cuda()
    dlo0 = torch.randn(N, C + 1, h, w, generator=g).cuda()
    sc, sh = (torch.rand(C + 2, generator=g) + 0.5).cuda(), (torch.randn(C + 2, generator=g) * 0.2).cuda()
    hsc, hsh = (torch.rand(C + 1, generator=g) + 0.5).cuda(), (torch.randn(C + 1, generator=g) * 0.2).cuda()

    def run_all():
        res = []
        for affine in (False, True):
            for with_hi in (True, False):
                out = torch.full((N, C + 1, H, W), 7.0, device="cuda")
                layers.upsample2x_add_fwd(lo, 1, C, out, 1, hi=hi if with_hi else None, hi_coff=1 if with_hi else 0, lo_relu=affine, hi_relu=with_hi and affine,
                                          lo_scale=sc[1:] if affine else None, lo_shift=sh[1:] if affine else None,
                                          hi_scale=hsc[1:] if affine and with_hi else None, hi_shift=hsh[1:] if affine and with_hi else None)
                res.append(out)
        for acc in (False, True):
            dlo = dlo0.clone()
            layers.upsample2x_bwd(do, 2, dlo, 1, C, accumulate=acc)
            res.append(dlo)
            dhi = hi.clone()
            layers.add_slice(do, 2, dhi, 1, C, accumulate=acc)
            res.append(dhi)
            dx = hi.clone()
            layers.avgpool2_bwd(lo, 1, dx, 1, C, accumulate=acc)
            res.append(dx)
        for affine in (False, True):
            y = torch.full((N, C + 2, h, w), 3.0, device="cuda")
            layers.avgpool2_fwd(hi, 1, C, y, 2, in_relu=affine, in_scale=hsc[1:] if affine else None, in_shift=hsh[1:] if affine else None)
            res.append(y)
        torch.cuda.synchronize()
        return res

    try:
        layers.set_layers_mode(1)
        want = run_all()
    finally:
        layers.set_layers_mode(0)
    got = run_all()
    assert len(got) == len(want) == 12
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, (a - b).abs().max().item())

