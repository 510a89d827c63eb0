#!/usr/bin/env python3
"""What would ONE launch per inception (heterogeneous workgroups co-resident on a CU) buy?  The three k x k branch convolutions
of an inception, forward shapes, (a) back to back on one stream, (b) on three streams at once.  GPU-bound queues of `reps`
repetitions, timed with events."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.ops import conv as C
from consistent_depth_amd import _native

dev = torch.device("cuda", 0)
N = 8
CASES = [("A2 384x224", 384, 224, 64, 16, (3, 7, 11)), ("B 192x112", 192, 112, 32, 32, (3, 5, 7)), ("C 192x112", 192, 112, 64, 32, (3, 7, 11)),
         ("B2 192x112", 192, 112, 64, 32, (3, 5, 7)), ("E 96x56", 96, 56, 32, 64, (3, 5, 7)), ("F 96x56", 96, 56, 64, 64, (3, 7, 11)),
         ("E 48x28", 48, 28, 32, 64, (3, 5, 7)), ("F 48x28", 48, 28, 64, 64, (3, 7, 11)), ("E 24x14", 24, 14, 32, 64, (3, 5, 7))]
reps = 20
for name, H, W, cin, cout, kss in CASES:
    P = torch.randn(N, 3 * cin + 3 * cout, H, W, device=dev)
    sc, sh = torch.rand(P.shape[1], device=dev) + 0.5, torch.randn(P.shape[1], device=dev) * 0.1
    stats = torch.zeros(_native.BN_STAT_SLOTS, P.shape[1], 2, dtype=torch.float64, device=dev)
    jobs = []
    for i, k in enumerate(kss):
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        pk = C.pack_weights(w)
        cfg = C.tuned_config(k, cin, cout, N, H, W, dev, affine_in=True, relu_in=True, stats=True, x_ctot=P.shape[1], y_ctot=P.shape[1])
        jobs.append((k, pk, i * cin, 3 * cin + i * cout, cfg))

    def run(job):
        k, pk, xo, yo, cfg = job
        C.conv2d(P, pk, cin, cout, k, x_coff=xo, out=P, y_coff=yo, in_scale=sc[xo:xo + cin], in_shift=sh[xo:xo + cin], in_relu=True,
                 stats=stats.view(-1), cfg=cfg)
    for j in jobs:
        run(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    single = []
    for j in jobs:
        e0.record()
        for _ in range(reps):
            run(j)
        e1.record(); torch.cuda.synchronize()
        single.append(e0.elapsed_time(e1) / reps * 1e3)
    e0.record()
    for _ in range(reps):
        for j in reversed(jobs):
            run(j)
    e1.record(); torch.cuda.synchronize()
    serial = e0.elapsed_time(e1) / reps * 1e3
    streams = [torch.cuda.Stream(device=dev) for _ in jobs]
    cur = torch.cuda.current_stream(dev)
    e0.record()
    for s in streams:
        s.wait_stream(cur)
    for _ in range(reps):
        for j, s in zip(reversed(jobs), streams):
            with torch.cuda.stream(s):
                run(j)
    for s in streams:
        cur.wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    conc = e0.elapsed_time(e1) / reps * 1e3
    # the same two schedules captured into HIP graphs (no host cost per launch)
    def capture(fn):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        return g

    def serial_fn():
        for _ in range(reps):
            for j in reversed(jobs):
                run(j)

    def conc_fn():
        cur = torch.cuda.current_stream(dev)
        for _ in range(reps):
            fork = torch.cuda.Event(); fork.record(cur)
            for j, s in zip(reversed(jobs), streams):
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    run(j)
                done = torch.cuda.Event(); done.record(s)
                cur.wait_event(done)
    res = []
    for fn in (serial_fn, conc_fn):
        g = capture(fn)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{name:12s} graph: serial {res[0]:7.1f} us  fork/join per inception {res[1]:7.1f} us ({res[0] / res[1]:.2f}x)")
    print(f"{name:12s} k={kss} each {[round(t, 1) for t in single]} us  serial {serial:7.1f}  3 streams {conc:7.1f} us  ({serial / conc:.2f}x)", flush=True)
