#!/usr/bin/env python3
"""Micro-benchmark of the fused loss kernel alone (no CNN): per-launch time from the library's
HIP-event hook, algorithmic GB/s (10*H*W*4 B per pair) vs the 8 TB/s peak.

    python tools/loss_bench.py --batches 4,32,256,1024 --iters 20 [--fwd-only]
Used under rocprofv3 for the kernel-trace / PMC evidence in profiles/.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4,32,256,1024")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=0, help="pairs per source+gather launch pair (0 = default)")
    ap.add_argument("--variant", type=int, default=0, help="0 = default dispatch, 4 = row sweep, 3 = evaluate-once + slab reduce")
    ap.add_argument("--pxt", type=int, default=0, help="row sweep: pixels per thread (0 = default rule)")
    ap.add_argument("--noise-px", type=float, default=0.25)
    ap.add_argument("--warm", type=int, default=3, help="untimed warm-up calls per batch size")
    ap.add_argument("--brief", action="store_true", help="one short line per batch size")
    ap.add_argument("--inconsistent", action="store_true", help="adversarial generator: unrelated depth per frame")
    args = ap.parse_args()
    from consistent_depth_amd import _native, synthetic
    from consistent_depth_amd.loss import consistency_loss as CL
    lib = _native.lib()
    assert lib.cd_debug_set_loss_variant(args.variant) == 0
    assert lib.cd_debug_set_loss_chunk(args.chunk) == 0
    assert lib.cd_debug_set_loss_sweep(args.pxt) == 0
    dev = torch.device("cuda", 0)
    H, W = args.height, args.width
    gen = synthetic.make_pair_batch if args.inconsistent else synthetic.make_scene_batch
    base = gen(8, H, W, seed=99, noise_px=args.noise_px)
    res = []
    for B in [int(b) for b in args.batches.split(",")]:
        rep = (B + 7) // 8
        t = lambda a: torch.tensor(a, device=dev).repeat((rep,) + (1,) * (a.ndim - 1))[:B].contiguous()  # noqa: E731
        depth = t(base["depth"])
        x = torch.log(depth) if args.mode == 1 else (1.0 / depth if args.mode == 2 else depth)
        x = (x + 0.01 * torch.randn_like(x)).requires_grad_(not args.fwd_only)
        flows, masks = [t(f) for f in base["flows"]], [t(m) for m in base["masks"]]
        intr, extr = t(base["intrinsics"]), t(base["extrinsics"])
        msum, twin = CL.mask_sums(masks[0], masks[1]), CL.tile_windows(flows, masks)  # dataset constants, cached
        call = lambda: CL.consistency_loss(x, flows, masks, intr, extr, 1.0, 0.1, mask_sums=msum, depth_mode=args.mode,  # noqa: E731
                                           tile_windows=None if args.fwd_only else twin)
        for _ in range(args.warm):
            call()
        torch.cuda.synchronize()
        assert lib.cd_profile_begin(args.iters) == 0
        for _ in range(args.iters):
            call()
        torch.cuda.synchronize()
        ms = (ctypes.c_float * args.iters)()
        bs = (ctypes.c_int * args.iters)()
        n = ctypes.c_int(0)
        assert lib.cd_profile_end(ms, bs, args.iters, ctypes.byref(n)) == 0
        ms = np.array(ms[:n.value])
        per_pair = (8 if args.fwd_only else 10) * H * W * 4
        gbs = per_pair * B / (ms * 1e-3) / 1e9
        res.append({"variant": args.variant, "pxt": args.pxt, "pairs": B, "avg_ms": float(ms.mean()), "min_ms": float(ms.min()),
                    "GBps_avg": float(per_pair * B / (ms.mean() * 1e-3) / 1e9), "GBps_best": float(gbs.max()),
                    "frac_of_8TBps": float(per_pair * B / (ms.mean() * 1e-3) / 1e9 / 8000.0)})
        res[-1]["ms_series"] = [round(float(x), 4) for x in ms]      # in call order (a clock ramp shows here)
        if args.brief:
            print(f"{B} avg {ms.mean():.4f} frac {res[-1]['frac_of_8TBps']:.4f} min {ms.min():.4f} med {np.median(ms):.4f} max {ms.max():.4f} "
                  f"first5 {np.round(ms[:5], 4).tolist()} last5 {np.round(ms[-5:], 4).tolist()}", flush=True)
        else:
            print(json.dumps(res[-1]), flush=True)
        del depth, x, flows, masks
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
