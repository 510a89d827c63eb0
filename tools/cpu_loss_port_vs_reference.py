#!/usr/bin/env python3
"""Build container only (needs /root/reference): how bench.py's `cpu_baseline` loss -- the C oracle, oracle/cd_oracle.c -- relates to the
reference's OWN code, `loss.consistency_loss.ConsistencyLoss` forward + autograd backward, on the same BS4 batch of 384x224 pairs.
bench.py cannot import the reference on the GPU box (it does not exist there), so its CPU leg is a "port"; this is the measured
relation between the two, committed as profiles/cpu_loss_port_vs_reference_r05.txt.

    python tools/cpu_loss_port_vs_reference.py [threads]
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import gen_golden as G, oracle   # noqa: E402
from consistent_depth_amd import synthetic   # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    ConsistencyLoss, _, _, _ = G._import_reference()
    batch = synthetic.make_scene_batch(4, 384, 224, seed=1)
    # the reference: forward + backward through autograd, float32 (its native dtype), `threads` torch threads
    ref_t = []
    for i in range(6):
        t0 = time.perf_counter()
        out = G._run_loss(ConsistencyLoss, batch, 1.0, 0.1, torch.float32)
        if i:
            ref_t.append(time.perf_counter() - t0)
    # the port: loss + analytic gradient, float32, single-threaded C
    port_t = []
    for i in range(11):
        t0 = time.perf_counter()
        o = oracle.consistency_loss(batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], 1.0, 0.1,
                                    dtype=np.float32)
        if i:
            port_t.append(time.perf_counter() - t0)
    rel = abs(float(out["total"]) - float(o["total"][0])) / abs(float(o["total"][0]))
    print(f"BS4 x 384x224, float32, build container ({os.cpu_count()} cores):")
    print(f"  reference  ConsistencyLoss forward + autograd backward, {threads} torch threads : median {1e3 * np.median(ref_t):8.1f} ms "
          f"(min {1e3 * min(ref_t):.1f}, max {1e3 * max(ref_t):.1f}; 5 calls after 1 warm-up) = {4 / np.median(ref_t):7.2f} pairs/s")
    print(f"  port       oracle/cd_oracle.c loss + analytic gradient, 1 thread                   : median {1e3 * np.median(port_t):8.1f} ms "
          f"(min {1e3 * min(port_t):.1f}, max {1e3 * max(port_t):.1f}; 10 calls after 1 warm-up) = {4 / np.median(port_t):7.2f} pairs/s")
    print(f"  port / reference time = {np.median(port_t) / np.median(ref_t):.3f};  |total_port - total_reference| / |total| = {rel:.2e}")


if __name__ == "__main__":
    main()
