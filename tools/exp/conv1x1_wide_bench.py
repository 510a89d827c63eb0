"""Dense 1x1 convolutions of MiDaS' ResNeXt-101 encoder at 384x384, 16 images: forward / input gradient (conv1x1_split_kc_kernel vs the staged
fp32 kernel, CD_AMD_CONV1X1_KC=0 in a second process) and weight gradient; us per launch and fp32-equivalent TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import conv

N = 16
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for (Cin, Cout, H, n) in [(256, 512, 48, 1), (512, 512, 48, 7), (512, 1024, 24, 1), (1024, 1024, 24, 45), (1024, 2048, 12, 1), (2048, 2048, 12, 5), (256, 256, 96, 5), (64, 256, 96, 2)]:
    x = torch.randn(N, Cin, H, H, device="cuda"); dy = torch.randn(N, Cout, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / Cin ** 0.5
    pk, pkT = conv.pack_weights(w), conv.pack_weights(w, transposed=True)
    y = torch.empty(N, Cout, H, H, device="cuda"); dx = torch.empty(N, Cin, H, H, device="cuda")
    dw = torch.empty(Cout, Cin, 1, 1, device="cuda"); ws = conv.wgrad_workspace(Cout, Cin, 1, "cuda")
    gf = 2.0 * N * H * H * Cin * Cout / 1e9
    tf = timed(lambda: conv.conv2d(x, pk, Cin, Cout, 1, out=y))
    td = timed(lambda: conv.conv2d(dy, pkT, Cout, Cin, 1, out=dx))
    tw = timed(lambda: conv.conv2d_wgrad(x, dy, Cin, Cout, 1, dw, ws))
    print(f"{Cin:4d}->{Cout:4d} @{H:2d}x{H:2d} x{n:2d}/net: fwd {tf:7.1f} us {gf / tf * 1e3:6.1f} TF/s | dgrad {td:7.1f} us {gf / td * 1e3:6.1f} | wgrad(+unpack) {tw:7.1f} us {gf / tw * 1e3:6.1f}   [{gf:.1f} GF]")
