#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Per-epoch parity at the HEADLINE shape: tests/golden/loop_16f_384x224.npz.

BASELINE.json asks for "depth maps and per-epoch losses matching the reference PyTorch CPU path within 1e-3 relative L1".  The
6-frame 64x48 loop golden (oracle/gen_golden_loop.py) pins the loop to the reference's own code; this file pins the NUMBERS at
384x224, where fp64 ground truth costs about a minute per training step (oracle/conv64.py) and cannot be computed inside a GPU test:

  stage 1  `snapshot <dir>`  (GPU box, through gpurun): the product's DepthFineTuner runs K burn-in epochs on the seeded 16-frame
           384x224 clip from the seeded random init (from a random init Adam's first steps are sign-like and amplify round-off --
           DESIGN.md section 2 -- so the 1e-3 budget is measured from a warm state); at the end of epoch K the weights, BatchNorm
           buffers, Adam moments and step count are written to <dir>/snap384.npz (64 MB: scratch, never committed).
  stage 2  `golden <dir>`    (build container, ~35 min of CPU): oracle/cpu_loop.py -- the reference's loop restated, pinned to
           /root/reference's own DepthFineTuner by tests/test_reference_loop_live_cpu.py -- continues FROM THAT SNAPSHOT in fp64 for
           T epochs over the same batches and writes every artefact of those epochs as the golden: eval/loss_e*.json (per pair and
           mean), eval/depth_*.raw, depth/frame_*.raw (every 2nd pixel in both directions: 1/4 of each map, to keep the fixture at
           a few MB), the final checkpoint (every 16th element of every tensor) and the snapshot's checksums.
  test     tests/test_loop_gpu.py::test_epochs_at_the_headline_shape_within_1e_3 re-runs the product from the seeds (the step is
           bit-reproducible: the regenerated snapshot is compared with the golden's checksums) and asserts <= 1e-3 on every
           artefact of epochs K+1 .. K+T.

    gpurun -- 'python -m oracle.gen_golden_loop_384 snapshot gpurun_out/snap384'
    python -m oracle.gen_golden_loop_384 golden gpurun_out/snap384
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

CLIP = dict(n_frames=16, H=384, W=224, seed=5)
K, T, INIT_SEED = 3, 2, 0          # burn-in epochs (10 steps each: 37 pairs, BS4), compared epochs
GOLDEN = os.path.join(REPO, "tests", "golden", "loop_16f_384x224.npz")
PX_STRIDE, CKPT_STRIDE = 2, 16


def initial_state():
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    torch.manual_seed(INIT_SEED)
    return HourglassModel().state_dict()


def run_product(work_dir):
    """The product loop on the GPU: K + T epochs; returns (fine-tuner, snapshot at the end of epoch K, {epoch: plan})."""
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.params import Video3dParamsParser
    path = os.path.join(work_dir, "clip")
    range_dir, pairs = msd.write_dataset(path, **CLIP)
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", str(K + T), "--batch_size", "4", "--print_freq", "0"])
    ft = DepthFineTuner(range_dir, list(range(CLIP["n_frames"])), params)
    ft.model.netG.load_state_dict(initial_state())
    snap, save = {}, ft.model.save

    def save_and_snapshot(file_name):      # end of every epoch (save_epoch_freq = 1), after its validation sweep
        save(file_name)
        if os.path.basename(file_name) == f"{K:04d}.pth":
            torch.cuda.synchronize()
            opt = getattr(ft._step, "step", ft._step).opt
            names = {id(p): n for n, p in ft.model.netG.named_parameters()}
            snap["state"] = {k: v.detach().cpu().clone() for k, v in ft.model.netG.state_dict().items()}
            snap["m1"] = {names[id(p)]: opt.exp_avg[o:o + p.numel()].detach().cpu().clone() for p, o in zip(opt._params, opt._offsets)}
            snap["m2"] = {names[id(p)]: opt.exp_avg_sq[o:o + p.numel()].detach().cpu().clone() for p, o in zip(opt._params, opt._offsets)}
            snap["k"] = int(opt.step_dev.item())
    ft.model.save = save_and_snapshot
    plans, orig = {}, ft.epoch_plan

    def recording_plan(epoch):
        plans.setdefault(epoch, orig(epoch))
        return plans[epoch]
    ft.epoch_plan = recording_plan
    ft.fine_tune()
    ft.save_depth()
    return ft, snap, plans, path, range_dir


def checksums(snap):
    """Order-independent fingerprints of a snapshot (fp64 sums of the fp32 values: exact enough to expose a single flipped bit)."""
    out = {}
    for part in ("state", "m1", "m2"):
        vals = [v.double() for v in snap[part].values() if v.is_floating_point()]
        out[part] = np.array([sum(float(v.sum()) for v in vals), sum(float(v.abs().sum()) for v in vals),
                              sum(float((v * v).sum()) for v in vals)], np.float64)
    return out


def subsample(maps):
    return np.ascontiguousarray(np.asarray(maps)[..., ::PX_STRIDE, ::PX_STRIDE])


def collect(out_dir, n_pairs):
    """Artefacts of epochs K+1 .. K+T of a finished run (product or CPU loop), in the golden's (sub-sampled) form."""
    from oracle import gen_golden_loop as G
    a = G.collect({"out_dir": out_dir, "steps": []})
    res = {}
    for e in range(K + 1, K + T + 1):
        tag = f"e{e:04d}_iter{e * n_pairs:06d}"
        for part in ("pairs", "reprojection", "disparity", "mean"):
            res[f"val_e{e}_{part}"] = a[f"val_{tag}_{part}"]
        res[f"evaldepth_e{e}"] = subsample(a[f"evaldepth_{tag}"])
    res["depth"] = subsample(a["depth"])
    sd = torch.load(os.path.join(out_dir, "checkpoints", f"{K + T:04d}.pth"), map_location="cpu")
    keys = [k for k in sd if sd[k].is_floating_point() and "uncertainty" not in k]
    res["ckpt_sample"] = np.concatenate([sd[k].double().reshape(-1)[::CKPT_STRIDE].numpy() for k in keys])
    res["num_batches_tracked"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")], np.int64)
    return res


def _pack(tensors):
    """{name: fp32 tensor} -> (byte planes of the concatenation, [(name, shape)]): the sign / exponent bytes of fp32 data compress,
    the mantissa bytes do not -- separated, the three 21 MB vectors of a snapshot fit gpurun's 64 MiB return channel."""
    flat = np.concatenate([v.numpy().astype(np.float32).ravel() for v in tensors.values()])
    return flat.view(np.uint8).reshape(-1, 4).T.copy(), [(k, list(v.shape)) for k, v in tensors.items()]


def _unpack(planes, table):
    flat = np.ascontiguousarray(planes.T).view(np.float32).ravel()
    out, o = {}, 0
    for name, shape in table:
        n = int(np.prod(shape)) if shape else 1
        out[name] = torch.as_tensor(flat[o:o + n].reshape(shape).copy())
        o += n
    return out


def stage_snapshot(dst):
    os.makedirs(dst, exist_ok=True)
    work = tempfile.mkdtemp(prefix="cd384_")
    ft, snap, plans, _, _ = run_product(work)
    assert snap and snap["k"] == K * len(plans[0]), (snap.get("k"), len(plans[0]))
    flat, tables = {}, {}
    for part in ("state", "m1", "m2"):
        fl = {k: v for k, v in snap[part].items() if v.is_floating_point()}
        flat[part], tables[part] = _pack(fl)
    flat["ints"] = np.array(json.dumps({k: int(v) for k, v in snap["state"].items() if not v.is_floating_point()}))
    flat["tables"] = np.array(json.dumps(tables))
    flat["k"] = np.array(snap["k"])
    store_pairs = [list(map(int, pr)) for pr in ft.store.pair_indices()]
    # the batches as PAIRS (frame, frame): independent of how a loader numbers its items
    flat["plans"] = np.array(json.dumps({str(e): [[store_pairs[i] for i in ids] for ids in p] for e, p in plans.items()}))
    for name, v in checksums(snap).items():
        flat["checksum_" + name] = v
    np.savez_compressed(os.path.join(dst, "snap384.npz"), **flat)
    prod = collect(ft.out_dir, len(ft.store))
    small = {k: v for k, v in prod.items() if not k.startswith("evaldepth") and k not in ("depth", "ckpt_sample")}
    np.savez_compressed(os.path.join(dst, "product384_small.npz"), **small)
    print("snapshot at step", snap["k"], "->", os.path.getsize(os.path.join(dst, "snap384.npz")) / 1e6, "MB;",
          {k: v.tolist() for k, v in checksums(snap).items()})
    for e in range(K + 1, K + T + 1):
        print(f"product epoch {e} mean", prod[f"val_e{e}_mean"].tolist())


def _cpu_run(src, dtype):
    """oracle/cpu_loop.py continued from the snapshot in `src` for T epochs in `dtype`; returns (artefacts, snapshot checksums, extras)."""
    import make_synthetic_dataset as msd
    from consistent_depth_amd.loaders.video_dataset import VideoDataset, load_color
    from oracle import conv64, cpu_loop
    conv64.ENABLED = True        # the dgemm formulation of the fp64 convolution: 5x faster on the build container's 8 cores
    z = np.load(os.path.join(src, "snap384.npz"))
    tables = json.loads(str(z["tables"]))
    snap = {part: _unpack(z[part], tables[part]) for part in ("state", "m1", "m2")}
    snap["k"] = int(z["k"])
    for name, v in json.loads(str(z["ints"])).items():
        snap["state"][name] = torch.tensor(v, dtype=torch.int64)
    plans = {int(e): p for e, p in json.loads(str(z["plans"])).items()}
    cs = checksums(snap)
    for name, v in cs.items():
        assert np.array_equal(v, z["checksum_" + name]), name
    work = tempfile.mkdtemp(prefix="cd384_")
    path = os.path.join(work, "clip")
    range_dir, pairs = msd.write_dataset(path, **CLIP)
    ds = VideoDataset(path, os.path.join(range_dir, "metadata_scaled.npz"))
    assert len(ds) == len(pairs)
    ds_idx = {tuple(int(v) for v in pr): i for i, pr in enumerate(ds.flow_indices)}
    plans = {e: [[ds_idx[tuple(pr)] for pr in batch] for batch in p] for e, p in plans.items()}
    out = os.path.join(work, "cpu")
    lp = cpu_loop.CpuLoop(ds, snap["state"], out, dtype=dtype)
    lp.ft.set_adam_state(snap["m1"], snap["m2"], snap["k"])
    lp.total_iters = K * len(ds)
    lp.fine_tune(T, lambda e: plans[K + e], start_epoch=K)
    lp.save_depth(out, list(range(CLIP["n_frames"])), lambda f: load_color(ds.color_fmt.format(f)))
    res = collect(out, len(ds))
    extras = {"clip": np.array([CLIP["n_frames"], CLIP["H"], CLIP["W"], CLIP["seed"]]), "K": np.array(K), "T": np.array(T),
              "k_steps": np.array(snap["k"]), "plans": z["plans"], "px_stride": np.array(PX_STRIDE), "ckpt_stride": np.array(CKPT_STRIDE),
              "pair_order": np.array([list(p) for p in ds.flow_indices], np.int64),
              "step_losses": np.array([l for _, _, l in lp.step_losses], np.float64)}
    for k in list(res):
        if res[k].dtype == np.float64 and res[k].size > 4096:
            res[k] = res[k].astype(np.float32)       # maps and the checkpoint sample: the artefacts themselves are fp32 files
    return res, cs, extras


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300))


def stage_golden(src):
    res, cs, extras = _cpu_run(src, torch.float64)
    res.update(extras)
    for name, v in cs.items():
        res["checksum_" + name] = v
    np.savez_compressed(GOLDEN, **res)
    print("wrote", GOLDEN, os.path.getsize(GOLDEN) / 1e6, "MB")
    for e in range(K + 1, K + T + 1):
        print(f"fp64 epoch {e} mean", res[f"val_e{e}_mean"].tolist())
    small = os.path.join(src, "product384_small.npz")
    if os.path.exists(small):
        p = np.load(small)
        for e in range(K + 1, K + T + 1):
            print(f"product (snapshot run) vs fp64, epoch {e} mean rel-L1: {_rel(p[f'val_e{e}_mean'], res[f'val_e{e}_mean']):.3e}")


def stage_ref32(src):
    """The YARDSTICK: the same continuation in the reference's own arithmetic (fp32 on the CPU, ~2 minutes) -- how far the reference
    is from its fp64 self on every artefact.  Stored next to the fp64 golden as `ref32dist_<artefact>` (distances only: the fp32
    artefacts themselves are one realisation of round-off and nothing is compared with them)."""
    res, cs, _ = _cpu_run(src, torch.float32)
    z = dict(np.load(GOLDEN))
    for name, v in cs.items():
        assert np.array_equal(v, z["checksum_" + name]), name
    for k, v in res.items():
        if v.dtype.kind == "f" and k in z:
            z["ref32dist_" + k] = np.array(_rel(v, z[k]))
            print(f"reference fp32 vs fp64  {k:24s} {float(z['ref32dist_' + k]):.3e}")
    np.savez_compressed(GOLDEN, **z)


if __name__ == "__main__":
    {"snapshot": stage_snapshot, "golden": stage_golden, "ref32": stage_ref32}[sys.argv[1]](sys.argv[2])
