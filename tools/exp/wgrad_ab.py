"""A/B of the weight-gradient kernel between two builds (CD_AMD_LIB): writes / compares dW for a list of shapes against an fp64 torch reference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import conv

SHAPES = [(8, 64, 64, 96, 56, 3), (8, 64, 32, 192, 112, 3), (8, 32, 64, 96, 56, 3), (8, 64, 16, 96, 56, 3), (4, 32, 32, 48, 28, 5),
          (2, 64, 64, 96, 56, 7), (2, 64, 64, 48, 28, 11), (8, 64, 64, 24, 14, 3), (3, 40, 48, 30, 24, 3)]
for (N, Cin, Cout, H, W, ks) in SHAPES:
    for affine in (False, True):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
        dy = torch.randn(N, Cout, H, W, device="cuda", generator=g)
        sc = torch.rand(Cin, device="cuda", generator=g) + 0.5 if affine else None
        sh = torch.randn(Cin, device="cuda", generator=g) * 0.3 if affine else None
        dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
        ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
        conv.conv2d_wgrad(x, dy, Cin, Cout, ks, dw, ws, in_scale=sc, in_shift=sh, in_relu=True)
        xa = x.double()
        if affine:
            xa = xa * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
        xa = xa.clamp_min(0)
        ref = torch.nn.grad.conv2d_weight(xa.cpu(), (Cout, Cin, ks, ks), dy.double().cpu(), padding=ks // 2)
        err = ((dw.double().cpu() - ref).abs().sum() / ref.abs().sum()).item()
        print(f"{(N, Cin, Cout, H, W, ks)} affine={affine}: rel-L1 {err:.3e}  max {((dw.double().cpu()-ref).abs().max()).item():.3e}", flush=True)
