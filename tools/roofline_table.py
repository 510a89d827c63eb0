#!/usr/bin/env python3
"""Roofline table of the convolutions from the committed sweeps (no GPU needed):

    python tools/roofline_table.py > profiles/conv_roofline_r01.txt
    python tools/roofline_table.py r02 > profiles/conv_roofline_r02.txt

r02: launches that run on the split-bf16 kernels (k >= 3 with >= 8 input channels; 1x1 forward / dgrad at >= 4096 row tiles,
> 16 output channels) are priced at the dense BF16 matrix peak divided by the six products per multiply-add
(2500 / 6 = 416.7 TFLOP/s of fp32-equivalent work); the others (1x1 weight gradient, RGB stem, small 1x1) at the fp32 peak.

For every forward / input-gradient / weight-gradient shape of one BS4 step (8 images): FLOPs, algorithmic HBM bytes
(read the two operands once, write the result once; weights are negligible), the two bounds at the MI355X peaks
(157.3 TFLOP/s fp32 MFMA, 8 TB/s HBM), the measured time of the best launch shape and the fraction of the binding roof."""
import json
import os

import sys

PEAK_TF, PEAK_BW = 157.3e12, 8.0e12
PEAK_SPLIT = 2500e12 / 6
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows():
    for line in open(os.path.join(ROOT, "profiles", f"conv_sweep_{TAG}.txt")):
        r = json.loads(line)
        if "shape" not in r:
            continue
        H, W, ks, ci, co, kind = r["shape"]
        px = 8 * H * W
        yield ("fwd" if kind == "f" else "dgrad", H, W, ks, ci, co, r["n"], r["best_us"], 2.0 * px * ci * ks * ks * co,
               4.0 * px * (ci + co) + (4.0 * px * co if kind == "d" else 0.0))      # dgrad accumulates: reads its output too
    for line in open(os.path.join(ROOT, "profiles", f"wgrad_sweep_{TAG}.txt")):
        r = json.loads(line)
        if "shape" not in r:
            continue
        H, W, ks, ci, co = r["shape"]
        px = 8 * H * W
        yield ("wgrad", H, W, ks, ci, co, r["n"], r["us"], 2.0 * px * ci * ks * ks * co, 4.0 * px * (ci + co))


def main():
    out, tot = [], {"t": 0.0, "mfma": 0.0, "hbm": 0.0, "roof": 0.0}
    for kind, H, W, ks, ci, co, n, us, flops, byts in rows():
        split = TAG != "r01" and ((ks >= 3 and ci >= 8) or (ks == 1 and kind != "wgrad" and co > 16 and 32 <= ci <= 384 and 8 * H * ((W + 31) // 32) >= 4096))
        t_mfma, t_hbm = flops / (PEAK_SPLIT if split else PEAK_TF) * 1e6, byts / PEAK_BW * 1e6
        roof = max(t_mfma, t_hbm)
        out.append((us * n, f"{kind:5s} {H:3d}x{W:<3d} k={ks:<2d} {ci:3d}->{co:<3d} x{n}  {flops / 1e9:8.2f} GFLOP {byts / 1e6:7.1f} MB  "
                            f"mfma {t_mfma:7.1f} us  hbm {t_hbm:6.1f} us  measured {us:7.1f} us  {'MFMA' if t_mfma >= t_hbm else 'HBM '}-bound: "
                            f"{100 * roof / us:5.1f} % of roof"))
        tot["t"] += us * n; tot["mfma"] += t_mfma * n; tot["hbm"] += t_hbm * n; tot["roof"] += roof * n
    print(__doc__.strip().split("\n\n")[1].replace("\n", " "))
    print()
    for _, line in sorted(out, reverse=True):
        print(line)
    print()
    print(f"per step: measured {tot['t'] / 1e3:.2f} ms; sum of MFMA bounds {tot['mfma'] / 1e3:.2f} ms, of HBM bounds {tot['hbm'] / 1e3:.2f} ms, "
          f"of the binding roofs {tot['roof'] / 1e3:.2f} ms -> {100 * tot['roof'] / tot['t']:.1f} % of roofline over all convolution launches")


if __name__ == "__main__":
    main()
