#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU's HBM, next to the loss call's bytes: the practical ceiling the roofline fraction
(algorithmic bytes / 8 TB/s) should be read against.  ATen element-wise kernels over planes of the loss call's size (256 pairs x 2 frames
x 384 x 224 floats = 176 MB per plane):
    read only   torch.sum(a)                       1 read
    copy        b.copy_(a)                         1 read + 1 write
    3 : 1       torch.addcmul(a, b, c, out=d)      3 reads + 1 write   (the loss call: 4 reads + 1 write per pixel)
HIP events around 50 calls after 20 warm-up calls; prints one JSON line."""
import json

import torch


def timed(fn, n=50, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda", 0)
    n = 256 * 2 * 384 * 224
    a, b, c, d = (torch.rand(n, device=dev) for _ in range(4))
    plane = n * 4
    res = {}
    for name, fn, planes in (("read_only_sum", lambda: torch.sum(a), 1), ("copy", lambda: b.copy_(a), 2),
                             ("addcmul_3r_1w", lambda: torch.addcmul(a, b, c, out=d), 4)):
        ms = timed(fn)
        res[name] = {"ms": round(ms, 4), "GBps": round(planes * plane / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(planes * plane / (ms * 1e-3) / 8e12, 4)}
    res["plane_MB"] = round(plane / 1e6, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
