#!/usr/bin/env python3
"""Where the HOST time of a graph-replayed fine-tuning step goes (bench.py's host_enqueue_ms_per_step read 5.6 ms on
the 244-frame clip and 14 ms on the 1000-frame clip against a 22 ms GPU step).

    python tools/host_cost.py [--frames 244|1000] [--steps 30]

Every step is timed piece by piece with the GPU IDLE at its start (synchronize before each step: nothing the host does can block on a full
queue or on the previous replay of the same graph), then the same loop is timed the way bench.py does (no synchronisation inside).  One
JSON line: per-piece host milliseconds (median), the free-running loop's host time per step, the GPU step time."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=244)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import bench
    from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    dev = torch.device("cuda", 0)
    model_cls = get_depth_model("mc")
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=model_cls.lambda_view_baseline, lambda_parameter=0,
                                learning_rate=model_cls.learning_rate, optimizer="Adam")
    model = model_cls(backend="hip", seed=0)
    model.train()
    step = GraphedFineTuneStep(FineTuneStep(model, params, world=1), eager_steps=2)
    gen = PairStore.synthetic_device if args.frames > 244 else PairStore.synthetic
    store = gen(args.frames, 384, 224, seed=0, device=dev)
    plans = bench.EpochPlans(len(store), 0, 1, 4, dev, seed=0)
    for _ in range(5):
        step.step_from_store(store, plans.next())
    torch.cuda.synchronize()
    assert step.graphed is True
    pieces = {k: [] for k in ("plan_next", "store_key", "gather_into", "replay", "post", "gpu_step")}
    from consistent_depth_amd import engine
    for _ in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = plans.next()
        t1 = time.perf_counter()
        skey = engine._store_key(store, ids)
        g = step._graphs[step._store_sig[skey]]
        t2 = time.perf_counter()
        store.gather_into(ids, g["images"], g["meta"])
        t3 = time.perf_counter()
        g["graph"].replay()
        t4 = time.perf_counter()
        step.step._weights_updated()
        out = g["guard"].clone(), {k: v.clone() for k, v in g["parts"].items()}
        t5 = time.perf_counter()
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        for k, v in zip(pieces, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t0)):
            pieces[k].append(1e3 * v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.step_from_store(store, plans.next())
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(json.dumps({"frames": args.frames, "pairs": len(store), "steps": args.steps,
                      "host_ms_gpu_idle_median": {k: round(float(np.median(v)), 4) for k, v in pieces.items()},
                      "host_ms_gpu_idle_max": {k: round(float(np.max(v)), 4) for k, v in pieces.items()},
                      "free_running_host_ms_per_step": round(1e3 * t_host / args.steps, 3),
                      "free_running_ms_per_step": round(1e3 * t_all / args.steps, 3)}))


if __name__ == "__main__":
    main()
