"""GPU parity of the fp32-MFMA convolution against torch.nn.functional.conv2d on the CPU
(fp64 reference; the kernel is an exact-fp32 fmaf chain, so the tolerance is fp32 round-off of a
K-long dot product)."""
import numpy as np
import pytest

from tests.gpu_util import report

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["split", "fp32", "split3"])
def arith(request):
    """Every test runs under the three arithmetic modes (cd_set_conv_arith): split-bf16 incl. the dedicated 1x1 kernel (default), the fp32
    matrix instruction, and split-bf16 for k >= 3 only."""
    from consistent_depth_amd import _native
    lib = _native.lib()
    before = lib.cd_get_conv_arith()
    assert lib.cd_set_conv_arith({"fp32": 0, "split3": 1, "split": 2}[request.param]) == 0
    yield request.param
    lib.cd_set_conv_arith(before)

CASES = [
    # N, Cin, Cout, H, W, ks
    (2, 3, 128, 24, 40, 7),      # stem shape (Cin not a multiple of the channel chunk)
    (2, 64, 16, 20, 33, 11),     # the dominant 11x11 64->16 (odd width)
    (1, 64, 32, 32, 48, 11),
    (2, 32, 32, 17, 31, 7),
    (2, 64, 64, 16, 32, 7),
    (2, 32, 64, 16, 24, 5),
    (2, 64, 32, 24, 32, 3),
    (2, 128, 64, 16, 40, 1),
    (2, 256, 32, 12, 16, 1),
    (1, 64, 1, 24, 40, 3),       # prediction head (Cout = 1)
    (1, 16, 64, 20, 36, 11),     # a dgrad-shaped case
]


def _ref(x, w, b, ks):
    import torch
    return torch.nn.functional.conv2d(x.double(), w.double(), b.double() if b is not None else None, padding=(ks - 1) // 2)


@pytest.mark.parametrize("ty", [0, 4, 8, 16])
@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", CASES)
def test_conv_fwd_matches_torch(N, Cin, Cout, H, W, ks, ty):
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    _native.lib().cd_debug_force_conv_tile_rows(ty)
    g = torch.Generator().manual_seed(ks * 1000 + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    ref = _ref(x, w, b, ks)
    pk = conv.pack_weights(w.cuda())
    try:
        y = conv.conv2d(x.cuda(), pk, Cin, Cout, ks, bias=b.cuda())
    finally:
        _native.lib().cd_debug_force_conv_tile_rows(0)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [(2, 64, 16, 40, 64, 11), (2, 16, 64, 40, 64, 11), (2, 64, 32, 32, 64, 7), (2, 32, 64, 24, 32, 5),
                                               (1, 64, 64, 96, 56, 11)])
def test_split_arithmetic_is_as_close_to_fp64_as_the_fp32_instruction(N, Cin, Cout, H, W, ks, arith):
    """The licence for the split-bf16 kernel: on the same inputs its distance to the fp64 result is not larger than that of
    the fp32 matrix instruction (both printed).  Inputs with a non-zero mean (post-ReLU activations) make the sums long."""
    if arith != "split":
        pytest.skip("one comparison covers both modes")
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    lib = _native.lib()
    g = torch.Generator().manual_seed(ks * 31 + Cin)
    x = torch.relu(torch.randn(N, Cin, H, W, generator=g) + 0.5)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
    ref = _ref(x, w, None, ks)
    pk = conv.pack_weights(w.cuda())
    err = {}
    for name, mode in (("split", 2), ("fp32", 0)):
        lib.cd_set_conv_arith(mode)
        y = conv.conv2d(x.cuda(), pk, Cin, Cout, ks)
        d = (y.cpu().double() - ref).abs()
        err[name] = (d.max().item() / ref.abs().max().item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    report("conv_split_vs_fp32", shape=f"{N}x{Cin}->{Cout}x{H}x{W}k{ks}", split_max=f"{err['split'][0]:.2e}", fp32_max=f"{err['fp32'][0]:.2e}",
           split_rms=f"{err['split'][1]:.2e}", fp32_rms=f"{err['fp32'][1]:.2e}")
    assert err["split"][0] <= 1.5 * err["fp32"][0] + 1e-8 and err["split"][1] <= 1.25 * err["fp32"][1] + 1e-9


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 128, 208, 128, 512), (1, 208, 128, 256, 500), (2, 100, 72, 128, 510), (4, 256, 160, 64, 512)])
def test_pointwise_convolution_at_dispatch_size(N, Cin, Cout, H, W, arith):
    """1x1 filters on images with >= 4096 row tiles: under "split" this is conv1x1_split.hip (filter slice resident in LDS,
    activations split in registers) -- channel slices of wider buffers, the fused input transform, the statistics epilogue and
    gradient accumulation, odd widths and channel counts; under the other modes the same call is the staged fp32 kernel."""
    import torch
    from consistent_depth_amd.ops import conv, layers
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin + 5, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda()
    base = torch.randn(N, Cout + 3, H, W, generator=g).cuda()
    pk = conv.pack_weights(w)
    act = torch.relu(x[:, 2:2 + Cin].double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = torch.einsum("nchw,oc->nohw", act, w.double().view(Cout, Cin)) + b.double().view(1, -1, 1, 1)
    out = base.clone()
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True, accumulate=True)
    got = (out[:, 1:1 + Cout] - base[:, 1:1 + Cout]).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert torch.equal(out[:, :1], base[:, :1]) and torch.equal(out[:, 1 + Cout:], base[:, 1 + Cout:])
    out2 = torch.full_like(base, 7.0)
    stats = layers.new_stats(Cout + 3, "cuda")
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out2, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True, stats=stats)
    err2 = (out2[:, 1:1 + Cout].double() - ref).abs().max().item() / ref.abs().max().item()
    st = stats.sum(0)
    report("conv_pointwise", arith=arith, shape=f"{N}x{Cin}->{Cout}x{H}x{W}", accumulate_err=f"{err:.2e}", plain_err=f"{err2:.2e}")
    assert err < 3e-6 and err2 < 2e-6
    assert (out2[:, :1] == 7).all() and (out2[:, 1 + Cout:] == 7).all()
    torch.testing.assert_close(st[1:1 + Cout, 0], ref.sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(st[1:1 + Cout, 1], (ref ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    assert (st[:1] == 0).all() and (st[1 + Cout:] == 0).all()
    # bit-reproducible
    out3 = torch.full_like(base, 7.0)
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out3, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True)
    assert torch.equal(out3, out2)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 1024, 1024, 24, 24), (3, 512, 512, 48, 48), (1, 2048, 2048, 12, 12), (2, 1024, 2048, 12, 12), (2, 256, 512, 48, 48),
                                            (1, 2048, 512, 6, 10), (2, 520, 136, 10, 10), (1, 72, 600, 14, 18), (2, 512, 256, 96, 96),
                                            (16, 1024, 1024, 24, 24), (16, 520, 1000, 24, 24)])   # (the last two: 36 pixel groups x 8 slices, two rounds on the chip)
def test_pointwise_convolution_with_wide_filters(N, Cin, Cout, H, W, arith):
    """Dense 1x1 filters with >= 512 channels on one side (the ResNeXt-101 encoder of MiDaS v2, BASELINE configs[4]): under "split" this is
    conv1x1_split.hip::conv1x1_split_kc_kernel (round 6: the filter slice streamed through LDS in double-buffered 64-channel chunks,
    pixel tiles over the flattened (image, y, x) index) -- the shapes of layer2 / 3 / 4 at 384x384, planes that are not a multiple of 32
    pixels, tiles that straddle two images, odd channel counts, channel slices of wider buffers, the fused input transform, the
    statistics epilogue and gradient accumulation; under the other modes the same call is the staged fp32 kernel (same bounds)."""
    import torch
    from consistent_depth_amd.ops import conv, layers
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin + 5, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda()
    base = torch.randn(N, Cout + 3, H, W, generator=g).cuda()
    pk = conv.pack_weights(w)
    act = torch.relu(x[:, 2:2 + Cin].double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    ref = torch.einsum("nchw,oc->nohw", act, w.double().view(Cout, Cin)) + b.double().view(1, -1, 1, 1)
    out = base.clone()
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True, accumulate=True)
    got = (out[:, 1:1 + Cout] - base[:, 1:1 + Cout]).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert torch.equal(out[:, :1], base[:, :1]) and torch.equal(out[:, 1 + Cout:], base[:, 1 + Cout:])
    out2 = torch.full_like(base, 7.0)
    stats = layers.new_stats(Cout + 3, "cuda")
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out2, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True, stats=stats)
    err2 = (out2[:, 1:1 + Cout].double() - ref).abs().max().item() / ref.abs().max().item()
    st = stats.sum(0)
    # plain input (no transform, no bias): the encoder's own call
    xin = x[:, 2:2 + Cin].contiguous()
    out4 = conv.conv2d(xin, pk, Cin, Cout, 1)
    ref4 = torch.einsum("nchw,oc->nohw", xin.double(), w.double().view(Cout, Cin))
    err4 = (out4.double() - ref4).abs().max().item() / ref4.abs().max().item()
    # the input gradient: the same kernel on the transposed packing
    pkT = conv.pack_weights(w, transposed=True)
    dy = torch.randn(N, Cout, H, W, generator=g).cuda()
    dx = conv.conv2d(dy, pkT, Cout, Cin, 1)
    refd = torch.einsum("nohw,oc->nchw", dy.double(), w.double().view(Cout, Cin))
    errd = (dx.double() - refd).abs().max().item() / refd.abs().max().item()
    report("conv_pointwise_wide", arith=arith, shape=f"{N}x{Cin}->{Cout}x{H}x{W}", accumulate_err=f"{err:.2e}", plain_err=f"{err2:.2e}", raw_err=f"{err4:.2e}",
           dgrad_err=f"{errd:.2e}")
    assert err < 3e-6 and err2 < 2e-6 and err4 < 2e-6 and errd < 2e-6
    assert (out2[:, :1] == 7).all() and (out2[:, 1 + Cout:] == 7).all()
    torch.testing.assert_close(st[1:1 + Cout, 0], ref.sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(st[1:1 + Cout, 1], (ref ** 2).sum((0, 2, 3)), rtol=1e-5, atol=1e-2)
    assert (st[:1] == 0).all() and (st[1 + Cout:] == 0).all()
    out3 = torch.full_like(base, 7.0)          # bit-reproducible
    conv.conv2d(x, pk, Cin, Cout, 1, bias=b, x_coff=2, out=out3, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True)
    assert torch.equal(out3, out2)


def test_conv_channel_slices_fused_input_and_stats():
    """Reads a channel slice, applies the producer's BN-apply+ReLU on load, writes into a slice of a
    concat buffer and accumulates the batch statistics of the raw output."""
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator().manual_seed(7)
    N, H, W, ks = 2, 20, 36, 5
    xbuf = torch.randn(N, 48, H, W, generator=g)
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    w = torch.randn(24, 32, ks, ks, generator=g) / np.sqrt(32 * ks * ks)
    b = torch.randn(24, generator=g)
    xin = torch.relu(xbuf[:, 8:40] * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref = _ref(xin, w, b, ks)
    out = torch.full((N, 40, H, W), 7.0).cuda()
    from consistent_depth_amd.ops import layers
    stats = layers.new_stats(40, "cuda")
    conv.conv2d(xbuf.cuda(), conv.pack_weights(w.cuda()), 32, 24, ks, bias=b.cuda(), x_coff=8, out=out, y_coff=10,
                in_scale=scale.cuda(), in_shift=shift.cuda(), in_relu=True, stats=stats)
    o = out.cpu()
    assert (o[:, :10] == 7).all() and (o[:, 34:] == 7).all()
    assert (o[:, 10:34].double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    s = stats.sum(0).cpu()   # the partial copies
    np.testing.assert_allclose(s[10:34, 0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[10:34, 1].numpy(), (ref ** 2).sum((0, 2, 3)).numpy(), rtol=1e-5)
    assert (s[:10] == 0).all() and (s[34:] == 0).all()


@pytest.mark.parametrize("Cin,Cout,ks", [(64, 16, 11), (32, 32, 7), (128, 64, 1), (3, 128, 7)])
def test_dgrad_is_conv_with_transposed_filter(Cin, Cout, ks):
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator().manual_seed(ks + Cin)
    N, H, W = 2, 18, 34
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, ks, ks, generator=g, dtype=torch.float64) / np.sqrt(Cin * ks * ks)
    dy = torch.randn(N, Cout, H, W, generator=g, dtype=torch.float64)
    torch.nn.functional.conv2d(x, w, padding=(ks - 1) // 2).backward(dy)
    pk = conv.pack_weights(w.float().cuda(), transposed=True)
    dx = conv.conv2d(dy.float().cuda(), pk, Cout, Cin, ks)
    assert (dx.cpu().double() - x.grad).abs().max().item() < 2e-5 * x.grad.abs().max().item()


@pytest.mark.parametrize("pipe", [1, 0])
@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [(2, 40, 56, 20, 36, 7), (2, 24, 64, 12, 20, 5), (2, 64, 48, 17, 33, 3),
                                               (2, 100, 64, 16, 40, 1), (1, 36, 30, 24, 36, 1), (1, 20, 16, 24, 40, 11),
                                               (2, 70, 208, 12, 40, 1), (1, 128, 112, 16, 36, 1), (2, 33, 256, 9, 35, 1)])
def test_launch_shapes_are_bit_identical(N, Cin, Cout, H, W, ks, pipe):
    """Every (tile rows, co tiles per workgroup) launch shape, with and without the register-prefetch pipeline, with
    the fused input transform, the statistics epilogue and gradient accumulation, produces the same bits (the order
    of accumulation over (channel chunk, tap) is fixed) -- the licence for timing-based autotuning."""
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv, layers
    lib = _native.lib()
    g = torch.Generator().manual_seed(ks * 77 + Cin)
    x = torch.randn(N, Cin + 5, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    sc, sh = (torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda()
    base = torch.randn(N, Cout + 3, H, W, generator=g).cuda()
    pk = conv.pack_weights(w)
    max_cot = lib.cd_conv2d_packed_co_tiles(Cout, ks)
    ref = _ref(torch.relu(x[:, 2:2 + Cin].cpu() * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)), w.cpu(), b.cpu(), ks)
    outs = []
    try:
        lib.cd_debug_set_conv_pipeline(pipe)
        for ty in (4, 8, 16, 32):
            for cot in (1, 2, 4, 8, 16):
                if cot > max_cot or (cot == 16 and ty > 4) or (cot == 8 and ty > 8) or (ty == 32 and cot > 1):
                    continue
                out = base.clone()
                stats = layers.new_stats(Cout + 3, "cuda")
                try:
                    conv.conv2d(x, pk, Cin, Cout, ks, bias=b, x_coff=2, out=out, y_coff=1, in_scale=sc, in_shift=sh, in_relu=True,
                                stats=stats, accumulate=True, cfg=(ty, cot))
                except RuntimeError:
                    if ty != 32:        # (hint 32 is a shape of the split-bf16 k x k kernels with > 16 output channels only: refused elsewhere)
                        raise
                    continue
                outs.append(((ty, cot), out, stats.sum(0)))
    finally:
        lib.cd_debug_set_conv_pipeline(1)
    (_, o0, s0) = outs[0]
    got = (o0[:, 1:1 + Cout] - base[:, 1:1 + Cout]).cpu().double()
    assert (got - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(o0[:, :1], base[:, :1]) and torch.equal(o0[:, 1 + Cout:], base[:, 1 + Cout:])
    for cfg, o, s in outs[1:]:
        assert torch.equal(o, o0), cfg
        torch.testing.assert_close(s, s0, rtol=1e-12, atol=1e-9)   # fp64 partials: only the order of fp64 additions differs


def test_autotuner_returns_a_valid_cached_launch_shape():
    from consistent_depth_amd.ops import conv
    cfg = conv.tuned_config(3, 32, 64, 2, 24, 32, "cuda", affine_in=True, relu_in=True, stats=True)
    assert cfg is not None and cfg[0] in (4, 8, 16, 32) and cfg[1] in (1, 2, 4)
    assert conv.tuned_config(3, 32, 64, 2, 24, 32, "cuda", affine_in=True, relu_in=True, stats=True) is cfg or \
        conv.tuned_config(3, 32, 64, 2, 24, 32, "cuda", affine_in=True, relu_in=True, stats=True) == cfg


@pytest.mark.parametrize("ks,Cin,Cout,N,H,W", [(3, 32, 64, 8, 96, 56), (1, 256, 208, 8, 96, 56), (7, 3, 128, 4, 96, 64), (11, 64, 16, 4, 96, 56)])
def test_weight_gradient_is_bit_reproducible(ks, Cin, Cout, N, H, W):
    """Every workgroup stores its partial sums into its own slice and the slices are added in split order: no atomics, one
    summation order -- two launches give the same bits (generic, wide-1x1 and few-input-channel plans), also through the
    deferred unpack table."""
    import torch
    from consistent_depth_amd.ops import conv
    torch.manual_seed(3)
    x, dy = torch.randn(N, Cin, H, W, device="cuda"), torch.randn(N, Cout, H, W, device="cuda")
    ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
    ws.fill_(float("nan"))        # the workspace needs no initialisation: every slice is written whole
    a = conv.conv2d_wgrad(x, dy, Cin, Cout, ks, torch.empty(Cout, Cin, ks, ks, device="cuda"), ws)
    ws2 = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
    ws2.fill_(7.0)
    b = conv.conv2d_wgrad(x, dy, Cin, Cout, ks, torch.empty(Cout, Cin, ks, ks, device="cuda"), ws2)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (Cout, Cin, ks, ks), dy.double().cpu(), padding=ks // 2)
    assert ((a.double().cpu() - ref).abs().sum() / ref.abs().sum()).item() < 2e-6
    g = torch.zeros(Cout, Cin, ks, ks, device="cuda")
    tab = conv.UnpackTable("cuda")
    conv.conv2d_wgrad(x, dy, Cin, Cout, ks, None, ws)
    tab.add(ws, lambda: g, Cin, ks, conv.wgrad_plan(Cout, Cin, ks, N, H, W))
    tab.run()
    assert torch.equal(g, a)


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks", [(8, 64, 64, 96, 56, 3), (8, 64, 32, 192, 112, 3), (8, 32, 64, 96, 56, 3), (8, 64, 16, 96, 56, 3),
                                               (4, 32, 32, 48, 28, 5), (2, 64, 64, 96, 56, 7), (2, 64, 64, 48, 28, 11), (3, 40, 48, 30, 24, 3)])
@pytest.mark.parametrize("mode", ["scale", "shift", "both"])
def test_weight_gradient_with_the_producers_affine(N, Cin, Cout, H, W, ks, mode):
    """dW of relu(x * scale + shift) -- the BatchNorm-apply-on-load form every weight gradient of the hourglass uses -- against fp64,
    for every block shape of the split-bf16 kernel (16 x 16 channels, the 32 x 16 block of the 3x3 gradient, 8-wave k = 11), scale and
    shift separately: round 3's first fetch-ahead build was exact without the affine and with a scale, and 1e-3 off -- a different
    amount on every run -- with a shift on the 32 x 16 block only (a packed-fma form the compiler chose there; wgrad_split.hip).
    Repeated launches must also agree bit for bit."""
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    dy = torch.randn(N, Cout, H, W, device="cuda", generator=g)
    sc = torch.rand(Cin, device="cuda", generator=g) + 0.5 if mode != "shift" else torch.ones(Cin, device="cuda")
    sh = torch.randn(Cin, device="cuda", generator=g) * 0.3 if mode != "scale" else torch.zeros(Cin, device="cuda")
    ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
    outs = [conv.conv2d_wgrad(x, dy, Cin, Cout, ks, torch.empty(Cout, Cin, ks, ks, device="cuda"), ws, in_scale=sc, in_shift=sh, in_relu=True).clone()
            for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    xa = (x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).clamp_min(0)
    ref = torch.nn.grad.conv2d_weight(xa.cpu(), (Cout, Cin, ks, ks), dy.double().cpu(), padding=ks // 2)
    err = ((outs[0].double().cpu() - ref).abs().sum() / ref.abs().sum()).item()
    assert err < 2e-6, err


def test_weight_gradients_of_many_convolutions_in_one_launch_per_class(arith):
    """cd_conv2d_wgrad_desc / cd_conv2d_wgrad_table (ops/conv.py::WgradTable): a mixed bag of the hourglass' k x k gradients -- large
    and tiny images, every kernel class, channel slices of shared buffers, with and without the producer's affine -- computed by one
    launch per class.  The packed partial sums must be BIT-identical to one cd_conv2d_wgrad per convolution (same workgroups, same
    tiles, same order), hence so are the unpacked gradients; gradients the table kernels do not cover (1x1, the RGB stem, everything
    under the fp32 arithmetic) are refused by add() and stay with the caller."""
    import torch
    from consistent_depth_amd.ops import conv
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(8, 32, 64, 24, 14, 3), (8, 32, 64, 24, 14, 5), (8, 32, 64, 24, 14, 7), (8, 64, 64, 48, 28, 11), (8, 64, 32, 96, 56, 3),
              (4, 64, 16, 96, 64, 11), (4, 64, 16, 96, 64, 7), (4, 32, 32, 48, 40, 5), (2, 64, 1, 40, 48, 3), (2, 48, 40, 20, 36, 7),
              (4, 128, 64, 24, 40, 1), (2, 3, 128, 24, 40, 7)]
    table = conv.WgradTable("cuda")
    jobs = []
    for i, (N, Cin, Cout, H, W, ks) in enumerate(shapes):
        X = torch.randn(N, Cin + 8, H, W, device="cuda", generator=g)          # the operands are channel slices of wider buffers
        DY = torch.randn(N, Cout + 5, H, W, device="cuda", generator=g)
        sc = torch.rand(Cin, device="cuda", generator=g) + 0.5 if i % 2 == 0 else None
        sh = torch.randn(Cin, device="cuda", generator=g) * 0.3 if i % 2 == 0 else None
        ws_a, ws_b = conv.wgrad_workspace(Cout, Cin, ks, "cuda"), conv.wgrad_workspace(Cout, Cin, ks, "cuda")
        ws_a.fill_(float("nan")); ws_b.fill_(float("nan"))
        kw = dict(x_coff=8, dy_coff=5, in_scale=sc, in_shift=sh, in_relu=i % 2 == 0)
        took = table.add(X, DY, Cin, Cout, ks, ws_b, **kw)
        assert took == (arith != "fp32" and ks >= 3 and Cin >= 8), (shapes[i], took)
        jobs.append((X, DY, Cin, Cout, ks, ws_a, ws_b, kw, took, (N, H, W)))
    if arith == "fp32":
        assert len(table) == 0
        return
    table.run()
    table.run()           # idempotent: every workgroup rewrites its whole slice
    for X, DY, Cin, Cout, ks, ws_a, ws_b, kw, took, (N, H, W) in jobs:
        if not took:
            continue
        conv.conv2d_wgrad(X, DY, Cin, Cout, ks, None, ws_a, **kw)             # deferred form: packed partial sums in ws_a
        cob, cib, splits = conv.wgrad_plan(Cout, Cin, ks, N, H, W)
        used = ((Cout + cob - 1) // cob) * ((Cin + cib - 1) // cib) * ks * ks * cob * cib * splits
        assert torch.equal(ws_a[:used].view(torch.int32), ws_b[:used].view(torch.int32)), (Cin, Cout, ks, H, W)
        assert torch.isnan(ws_b[used:]).all()                                   # nothing written beyond the plan's slices


@pytest.mark.parametrize("N,H,W,cin,cout,kss", [(8, 24, 14, 32, 64, (7, 5, 3)), (8, 48, 28, 64, 64, (11, 7, 3)), (4, 96, 56, 32, 32, (7, 5, 3)),
                                                 (2, 40, 72, 64, 32, (11, 7, 3)), (2, 33, 47, 24, 40, (5, 3)), (2, 40, 72, 64, 16, (11, 7, 3)), (4, 48, 64, 32, 16, (11, 7, 3))])
@pytest.mark.parametrize("cfg", [(4, 1), (8, 1), (16, 1), (4, 2), (16, 2), (32, 1)])
def test_branches_of_an_inception_in_one_dispatch(arith, N, H, W, cin, cout, kss, cfg):
    """cd_conv2d_fwd_multi: the k x k branches of an inception (different filter sizes, different input slices of ONE buffer, adjacent
    output slices, producer's BatchNorm applied on load, batch statistics in the epilogue) in one dispatch vs one launch each -- outputs
    and statistics bit for bit, for every shared launch shape; and the input-gradient form (transposed packs, accumulate)."""
    import torch
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    if cfg[0] == 32 and (arith == "fp32" or cout <= 16):
        pytest.skip("launch shape 32 (8 row tiles + two chunks per round): split-bf16 kernels with 32 output channels per column tile only")
    g = torch.Generator(device="cuda").manual_seed(11)
    nb = len(kss)
    P = torch.randn(N, nb * cin + nb * cout, H, W, device="cuda", generator=g)
    sc, sh = torch.rand(nb * cin, device="cuda", generator=g) + 0.5, torch.randn(nb * cin, device="cuda", generator=g) * 0.2
    ws = [torch.randn(cout, cin, k, k, device="cuda", generator=g) * 0.05 for k in kss]
    bs = [torch.randn(cout, device="cuda", generator=g) for _ in kss]

    def members(out, stats):
        return [dict(x=P, packed_w=conv.pack_weights(w), Cin=cin, Cout=cout, ks=k, bias=b, x_coff=i * cin, out=out, y_coff=nb * cin + i * cout,
                     in_scale=sc[i * cin:(i + 1) * cin], in_shift=sh[i * cin:(i + 1) * cin], in_relu=True, stats=stats)
                for i, (k, w, b) in enumerate(zip(kss, ws, bs))]
    outs, stats = [], []
    for merged in (False, True):
        out = torch.zeros_like(P)
        st = torch.zeros(_native.BN_STAT_SLOTS, P.shape[1], 2, dtype=torch.float64, device="cuda")
        ms = members(out, st.view(-1))
        if merged:
            took = conv.conv2d_multi(ms, cfg)
            assert took == (arith != "fp32")
            if not took:
                return
        else:
            for m in ms:
                conv.conv2d(m["x"], m["packed_w"], cin, cout, m["ks"], bias=m["bias"], x_coff=m["x_coff"], out=out, y_coff=m["y_coff"],
                            in_scale=m["in_scale"], in_shift=m["in_shift"], in_relu=True, stats=m["stats"], cfg=cfg)
        outs.append(out)
        stats.append(st)
    assert torch.equal(outs[0], outs[1])
    # (the statistics are fp64 atomics into 16 slots: sums of the same per-workgroup partials in a run-dependent order)
    assert torch.allclose(stats[0].sum(0), stats[1].sum(0), rtol=1e-12, atol=1e-9)
    # input gradient: dY = the branch outputs' slices -> the mid slices, accumulated on top of what is there
    base = torch.randn(N, P.shape[1], H, W, device="cuda", generator=g)
    res = []
    for merged in (False, True):
        dst = base.clone()
        ms = [dict(x=outs[0], packed_w=conv.pack_weights(w, transposed=True), Cin=cout, Cout=cin, ks=k, x_coff=nb * cin + i * cout, out=dst, y_coff=i * cin,
                   accumulate=True) for i, (k, w) in enumerate(zip(kss, ws))]
        if merged:
            assert conv.conv2d_multi(ms, cfg)
        else:
            for m in ms:
                conv.conv2d(m["x"], m["packed_w"], cout, cin, m["ks"], x_coff=m["x_coff"], out=dst, y_coff=m["y_coff"], accumulate=True, cfg=cfg)
        res.append(dst)
    assert torch.equal(res[0], res[1])
