import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from consistent_depth_amd.ops import conv
N = 8
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (H, W, ks, Cin, Cout) in [(384, 224, 1, 128, 208), (384, 224, 1, 128, 192), (192, 112, 1, 128, 128), (384, 224, 3, 64, 16), (192, 112, 3, 32, 32)]:
    x = torch.randn(N, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, ks, ks, device="cuda") * 0.05
    pk = conv.pack_weights(w); out = torch.empty(N, Cout, H, W, device="cuda")
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
    st = torch.zeros(16, Cout, 2, dtype=torch.float64, device="cuda")
    for cfg in [(8, 4), (8, 2), (16, 1), (16,2)]:
        if ks == 3 and cfg[1] > (1 if Cout <= 16 else 2): continue
        r = {}
        r["plain"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, cfg=cfg))
        r["relu"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, in_relu=True, cfg=cfg))
        r["affine+relu"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, in_scale=sc, in_shift=sh, in_relu=True, cfg=cfg))
        r["stats"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, stats=st, cfg=cfg))
        r["relu+stats"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, in_relu=True, stats=st, cfg=cfg))
        r["accumulate"] = t(lambda: conv.conv2d(x, pk, Cin, Cout, ks, out=out, accumulate=True, cfg=cfg))
        print((H, W, ks, Cin, Cout), cfg, {k: round(v, 1) for k, v in r.items()}, flush=True)
