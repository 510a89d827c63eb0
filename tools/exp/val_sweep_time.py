#!/usr/bin/env python3
"""Wall time of ONE validation sweep (eval_and_save: 715 pairs of 384x224 in 179 batches, train-mode BatchNorm forward + forward-only
loss + first-sighting depth export through the asynchronous writer) and of one training epoch, HIP-graph replay on / off."""
import argparse, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
from consistent_depth_amd.loaders.pair_store import PairStore
from consistent_depth_amd.params import Video3dParamsParser

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 244
tmp = tempfile.mkdtemp()
params = Video3dParamsParser().parse(["--path", tmp, "--batch_size", "4", "--print_freq", "0"])
store = PairStore.synthetic(frames, 384, 224, seed=0, device=torch.device("cuda", 0))
for graph in ("1", "0"):
    os.environ["CD_AMD_EVAL_GRAPH"] = graph
    ft = DepthFineTuner(os.path.join(tmp, "r" + graph), list(range(frames)), params, store=store)
    os.makedirs(os.path.join(ft.out_dir, "eval"), exist_ok=True)
    ft.model.train()
    step = FineTuneStep(ft.model, params, world=1)
    ft.eval_and_save(step, "_warm")            # plans, launch shapes, (graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ft.eval_and_save(step, "_t")
    torch.cuda.synchronize()
    print(f"validation sweep, {len(store)} pairs, eval graph {'on' if graph == '1' else 'off'}: {time.perf_counter() - t0:.3f} s "
          f"(capture error: {step._evaluator.capture_error})", flush=True)
