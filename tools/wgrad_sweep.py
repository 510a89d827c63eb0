#!/usr/bin/env python3
"""Weight-gradient kernel on every distinct convolution shape of one mc-hourglass step (8 images), weighted by
launches per step; with --parts also with the flush / the MFMAs switched off (cd_debug_set_wgrad_mode) to show
where the time goes."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.conv_sweep import step_shapes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--parts", action="store_true")
    args = ap.parse_args()
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    lib = _native.lib()
    N, tot = 8, 0.0
    rows = []
    for (H, W, ks, Cin, Cout, kind), cnt in step_shapes(args.height, args.width).items():
        if kind != "f":
            continue
        x = torch.randn(N, Cin, H, W, device="cuda"); dy = torch.randn(N, Cout, H, W, device="cuda")
        dw = torch.empty(Cout, Cin, ks, ks, device="cuda")
        ws = conv.wgrad_workspace(Cout, Cin, ks, "cuda")
        flops = 2.0 * N * H * W * Cin * ks * ks * Cout

        def timeit():
            conv.conv2d_wgrad(x, dy, Cin, Cout, ks, dw, ws, in_relu=True); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                conv.conv2d_wgrad(x, dy, Cin, Cout, ks, dw, ws, in_relu=True)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3
        rec = {"shape": [H, W, ks, Cin, Cout], "n": cnt, "us": round(timeit(), 1)}
        rec["TFLOPs"] = round(flops / rec["us"] / 1e6, 1)
        if args.parts:
            for name, bits in (("no_flush", 1), ("no_mfma", 2), ("staging_only", 3)):
                lib.cd_debug_set_wgrad_mode(bits)
                rec[name] = round(timeit(), 1)
            lib.cd_debug_set_wgrad_mode(0)
        tot += rec["us"] * cnt
        rows.append(rec)
    for rec in sorted(rows, key=lambda r: -r["us"] * r["n"]):
        print(json.dumps(rec), flush=True)
    print(json.dumps({"per_step_ms": round(tot / 1e3, 3)}))


if __name__ == "__main__":
    main()
