#!/usr/bin/env python3
"""Sweep (tile rows x co tiles per block x pipeline) of the MFMA convolution over EVERY distinct forward / dgrad
convolution shape of one mc-hourglass fine-tuning step (8 images), weighted by launches per step.

    python tools/conv_sweep.py [--height 384 --width 224] [--iters 5] > gpurun_out/conv_sweep.jsonl

Each line: shape, launches per step, the heuristic's choice and time, and the time of every configuration.
The last line sums the per-step time under the heuristic and under the per-shape best.
"""
import argparse
import json
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# inception specs / tree of the hourglass, as in oracle/hourglass_ref.py
SPEC = {
    "A": [[16], [3, 32, 16], [7, 32, 16], [11, 32, 16]], "A2": [[16], [3, 64, 16], [7, 64, 16], [11, 64, 16]],
    "B": [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]], "B2": [[32], [3, 64, 32], [5, 64, 32], [7, 64, 32]],
    "C": [[32], [3, 64, 32], [7, 64, 32], [11, 64, 32]], "D": [[64], [3, 32, 64], [5, 32, 64], [7, 32, 64]],
    "E": [[64], [3, 32, 64], [5, 32, 64], [7, 32, 64]], "F": [[64], [3, 64, 64], [7, 64, 64], [11, 64, 64]],
    "G": [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]],
}
CIN = {"A": 128, "A2": 128, "B": 128, "B2": 128, "C": 128, "D": 128, "E": 256, "F": 256, "G": 256}
LEVELS = {0: ["A2"], 1: ["B", "B", "B", "C", "B2", "A"], 2: ["B", "D", "E", "F", "E", "G"], 3: ["E"] * 5 + ["F"], 4: ["E"] * 3}


def step_shapes(H, W):
    """Counter of (H, W, ks, Cin, Cout, kind) over one step; kind 'f' = forward, 'd' = input gradient."""
    c = Counter()
    c[(H, W, 7, 3, 128, "f")] += 1            # stem (its input gradient is not needed)
    c[(H, W, 3, 64, 1, "f")] += 1             # prediction head
    c[(H, W, 3, 1, 64, "d")] += 1
    for lvl, names in LEVELS.items():
        h, w = H >> lvl, W >> lvl
        for nm in names:
            spec, cin = SPEC[nm], CIN[nm]
            entry = spec[0][0] + sum(mid for _, mid, _ in spec[1:])
            c[(h, w, 1, cin, entry, "f")] += 1     # fused 1x1 entry convolutions of the 4 branches
            c[(h, w, 1, entry, cin, "d")] += 1
            for k, mid, out in spec[1:]:
                c[(h, w, k, mid, out, "f")] += 1
                c[(h, w, k, out, mid, "d")] += 1
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--levels", default="0,1,2,3,4")
    ap.add_argument("--no-pipe-axis", action="store_true", help="only sweep with the pipeline on")
    ap.add_argument("--heur-only", action="store_true", help="time the dispatcher's own choice only (A/B of two builds: CD_AMD_LIB)")
    args = ap.parse_args()
    from consistent_depth_amd import _native
    from consistent_depth_amd.ops import conv
    lib = _native.lib()
    N = 8
    levels = {int(v) for v in args.levels.split(",")}
    tot_h = tot_b = 0.0
    for (H, W, ks, Cin, Cout, kind), cnt in sorted(step_shapes(args.height, args.width).items(), key=lambda kv: (-kv[0][0], kv[0][2:])):
        if (args.height // H).bit_length() - 1 not in levels:
            continue
        x = torch.randn(N, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, ks, ks, device="cuda") * 0.05
        pk = conv.pack_weights(w)
        out = torch.empty(N, Cout, H, W, device="cuda")
        sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.1
        stats = torch.zeros(16, Cout, 2, dtype=torch.float64, device="cuda")   # (CD_BN_STAT_SLOTS, Cout, 2)
        fused = kind == "f" and Cin > 3     # forward convs read BN-normalised inputs and produce batch statistics
        flops = 2.0 * N * H * W * Cin * ks * ks * Cout

        def run():
            if fused:
                conv.conv2d(x, pk, Cin, Cout, ks, out=out, in_scale=sc, in_shift=sh, in_relu=True, stats=stats)
            else:
                conv.conv2d(x, pk, Cin, Cout, ks, out=out, accumulate=(kind == "d"))

        def timeit():
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3   # us

        lib.cd_debug_force_conv_tile_rows(0); lib.cd_debug_force_conv_co_tiles(0); lib.cd_debug_set_conv_pipeline(1)
        t_h = timeit()
        if args.heur_only:
            tot_h += t_h * cnt
            print(json.dumps({"shape": [H, W, ks, Cin, Cout, kind], "n": cnt, "heur_us": round(t_h, 1), "TFLOPs": round(flops / t_h / 1e6, 1)}), flush=True)
            continue
        pack_cot = lib.cd_conv2d_packed_co_tiles(Cout, ks)
        res = {}
        for ty in (4, 8, 16):
            for cot in (1, 2, 4, 8, 16):
                if cot > pack_cot or (cot == 16 and ty > 4) or (cot == 8 and ty > 8):
                    continue
                for pipe in ((1,) if args.no_pipe_axis else (1, 0)):
                    lib.cd_debug_force_conv_tile_rows(ty); lib.cd_debug_force_conv_co_tiles(cot); lib.cd_debug_set_conv_pipeline(pipe)
                    try:
                        res[f"ty{ty}_co{cot}_p{pipe}"] = round(timeit(), 1)
                    except RuntimeError:
                        res[f"ty{ty}_co{cot}_p{pipe}"] = None
        lib.cd_debug_force_conv_tile_rows(0); lib.cd_debug_force_conv_co_tiles(0); lib.cd_debug_set_conv_pipeline(1)
        ok = {k: v for k, v in res.items() if v}
        best = min(ok, key=ok.get)
        tot_h += t_h * cnt
        tot_b += ok[best] * cnt
        print(json.dumps({"shape": [H, W, ks, Cin, Cout, kind], "n": cnt, "heur_us": round(t_h, 1), "best": best, "best_us": ok[best],
                          "best_TFLOPs": round(flops / ok[best] / 1e6, 1), "all": res}), flush=True)
    print(json.dumps({"per_step_ms_heuristic": round(tot_h / 1e3, 3), **({} if args.heur_only else {"per_step_ms_best": round(tot_b / 1e3, 3)})}))


if __name__ == "__main__":
    main()
