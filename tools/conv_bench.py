#!/usr/bin/env python3
"""Per-shape throughput of the MFMA convolution on the hourglass's dominant shapes (batch of 8 images)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # H, W, ks, Cin, Cout, share of forward MACs
    (384, 224, 11, 64, 16, 20.2), (192, 112, 11, 64, 32, 10.1), (192, 112, 7, 64, 32, 8.2), (384, 224, 7, 64, 16, 8.2),
    (192, 112, 7, 32, 32, 6.1), (96, 56, 11, 64, 64, 5.0), (384, 224, 1, 128, 64, 4.0), (192, 112, 5, 32, 32, 3.1),
    (384, 224, 7, 3, 128, 3.1), (96, 56, 7, 32, 64, 3.1), (192, 112, 1, 128, 32, 2.8), (192, 112, 3, 64, 32, 1.5),
    (48, 28, 7, 32, 64, 1.3), (24, 14, 7, 32, 64, 0.1), (384, 224, 3, 64, 16, 0.8), (96, 56, 3, 64, 64, 0.4), (384, 224, 1, 128, 208, 4.0), (192, 112, 1, 128, 224, 2.8), (96, 56, 1, 256, 160, 1.0),
]
DGRAD_SHAPES = [  # the input-gradient convolutions (channels swapped) of the dominant shapes
    (384, 224, 11, 16, 64, 20.2), (192, 112, 11, 32, 64, 10.1), (192, 112, 7, 32, 64, 8.2), (384, 224, 7, 16, 64, 8.2),
    (192, 112, 7, 32, 32, 6.1), (96, 56, 11, 64, 64, 5.0), (192, 112, 5, 32, 32, 3.1), (96, 56, 7, 64, 32, 3.1),
    (192, 112, 3, 32, 64, 1.5), (384, 224, 1, 208, 128, 4.6), (192, 112, 1, 128, 128, 2.8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--torch", action="store_true", help="also time torch (MIOpen) on the same shape")
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--dgrad", action="store_true", help="bench the dgrad-shaped convolutions instead")
    ap.add_argument("--wgrad", action="store_true", help="bench the weight-gradient kernel on the forward shapes")
    ap.add_argument("--arith", choices=["split", "fp32"], default=None, help="arithmetic of the k >= 5 convolutions (default: library default)")
    ap.add_argument("--cfgs", default="", help="launch shapes to time, e.g. '16x1,8x1,16x2' (tile rows x co tiles); default: the heuristic")
    args = ap.parse_args()
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    if args.arith:
        _native.lib().cd_set_conv_arith(1 if args.arith == "split" else 0)
    cfgs = [tuple(int(v) for v in c.split("x")) for c in args.cfgs.split(",") if c] or [None]
    N = 8
    for i, (H, W, ks, Cin, Cout, share) in enumerate(DGRAD_SHAPES if args.dgrad else SHAPES):
        if args.only >= 0 and i != args.only:
            continue
        x = torch.randn(N, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, ks, ks, device="cuda") * 0.05
        b = torch.zeros(Cout, device="cuda")
        pk = conv.pack_weights(w)
        out = torch.empty(N, Cout, H, W, device="cuda")
        flops = 2.0 * N * H * W * Cin * ks * ks * Cout

        def timeit(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters

        if args.wgrad:
            dy = torch.randn(N, Cout, H, W, device="cuda")
            dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
            ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
            ms = timeit(lambda: conv.conv2d_wgrad(x, dy, Cin, Cout, ks, dw, ws, in_relu=True))
        else:
            per_cfg = {("heuristic" if c is None else "%dx%d" % c): timeit(lambda: conv.conv2d(x, pk, Cin, Cout, ks, bias=b, out=out, cfg=c)) for c in cfgs}
            ms = min(per_cfg.values())
        rec = {"shape": [H, W, ks, Cin, Cout], "share_pct": share, "ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1)}
        if not args.wgrad and len(cfgs) > 1:
            rec["cfgs_ms"] = {k: round(v, 4) for k, v in per_cfg.items()}
        if args.torch:
            ms_t = timeit(lambda: torch.nn.functional.conv2d(x, w, b, padding=(ks - 1) // 2))
            rec["torch_ms"] = round(ms_t, 4)
            rec["torch_TFLOPs"] = round(flops / ms_t / 1e9, 1)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
