import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import conv
N, Cin, Cout, H, W, ks = 8, 64, 64, 96, 56, 3
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
dy = torch.randn(N, Cout, H, W, device="cuda", generator=g)
for mode in ("scale_only", "shift_only", "both"):
    sc = torch.rand(Cin, device="cuda", generator=g) + 0.5 if mode != "shift_only" else torch.ones(Cin, device="cuda")
    sh = torch.randn(Cin, device="cuda", generator=g) * 0.3 if mode != "scale_only" else torch.zeros(Cin, device="cuda")
    dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
    ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
    conv.conv2d_wgrad(x, dy, Cin, Cout, ks, dw, ws, in_scale=sc, in_shift=sh, in_relu=True)
    xa = (x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).clamp_min(0)
    ref = torch.nn.grad.conv2d_weight(xa.cpu(), (Cout, Cin, ks, ks), dy.double().cpu(), padding=ks // 2)
    err = (dw.double().cpu() - ref).abs()
    bad = (err > 1e-2).nonzero()
    print(mode, "bad entries", len(bad), "of", err.numel())
    print("  co:", sorted(set(bad[:, 0].tolist()))[:70])
    print("  ci:", sorted(set(bad[:, 1].tolist()))[:70])
    print("  ky:", sorted(set(bad[:, 2].tolist())), "kx:", sorted(set(bad[:, 3].tolist())))
    print("  first:", [(tuple(b.tolist()), round(err[tuple(b.tolist())].item(), 3), round(ref[tuple(b.tolist())].item(), 3)) for b in bad[:6]])
