#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Per-epoch parity at the HEADLINE shape over a FULL-LENGTH run (num_epochs = 20, the reference's
default, /root/reference/depth_fine_tuning.py:52): tests/golden/loop_16f_384x224.npz and tests/golden/loop_dense8f_384x224.npz.

BASELINE.json asks for "depth maps and per-epoch losses matching the reference PyTorch CPU path within 1e-3 relative L1".  The
6-frame 64x48 loop golden (oracle/gen_golden_loop.py) pins the loop to the reference's own code; this file pins the NUMBERS at
384x224, where fp64 ground truth costs about a minute per training step (oracle/conv64.py) and cannot be computed inside a GPU test.
Two clips (SPECS):

  "a"      16 frames, 37 pairs, BS4: 10 steps per epoch; masks keep 70 % of the in-bounds pixels (the clip of every earlier golden).
           K = 3 burn-in epochs, then T = 20 compared epochs = 200 steps + 20 validation sweeps + the export.
  "dense"  8 frames, 15 pairs, 4 steps per epoch; small camera motion and mask_keep = 1: every pair's masks cover >= 95 % of the
           pixels, i.e. (nearly) every pixel of every depth map is constrained by a valid flow.  K = 3, T = 10.  It answers whether the
           depth-map drift seen on clip "a" belongs to pixels no valid flow constrains.

  stage 1  `snapshot <spec> <dir>`  (GPU box, through gpurun): the product's DepthFineTuner runs K + T epochs on the seeded clip from
           the seeded random init (from a random init Adam's first steps are sign-like and amplify round-off -- DESIGN.md section 2 --
           so the 1e-3 budget is measured from a warm state); at the end of epoch K the weights, BatchNorm buffers, Adam moments and
           step count are written to <dir>/snap_<spec>.npz (64 MB: scratch, never committed), the product's own artefacts of the
           compared epochs to <dir>/product_<spec>.npz (for the curves of profiles/parity_20ep_r05.txt).
  stage 2  `golden <spec> <dir>`    (build container, HOURS of CPU): oracle/cpu_loop.py -- the reference's loop restated, pinned to
           /root/reference's own DepthFineTuner by tests/test_reference_loop_live_cpu.py -- continues FROM THAT SNAPSHOT in fp64 for
           T epochs over the same batches and writes the artefacts of EVERY epoch as the golden: eval/loss_e*.json (per pair and
           mean), eval/depth_*.raw (every 8th pixel in both directions), the checkpoint (every 256th element per epoch, every 16th
           of the last one), depth/frame_*.raw (every 2nd pixel) and the snapshot's checksums.  The work directory is persistent
           (<dir>/cpu_<spec>_<dtype>): `golden` on a directory whose run was interrupted collects the epochs that are complete.
  stage 3  `ref32 <spec> <dir>`     the same continuation in the reference's OWN arithmetic (fp32 on the CPU): its distances to the
           fp64 run are the yardstick stored next to each artefact (`ref32dist_*`).
  test     tests/test_loop_gpu.py::test_full_length_run_vs_fp64_and_vs_the_reference_fp32_run[...] re-runs the product from the seeds (the step is
           bit-reproducible: the regenerated snapshot is compared with the golden's checksums) and asserts the bounds at EVERY epoch.

    gpurun -- 'python -m oracle.gen_golden_loop_384 snapshot a gpurun_out/snap384'
    python -m oracle.gen_golden_loop_384 golden a gpurun_out/snap384
    python -m oracle.gen_golden_loop_384 ref32 a gpurun_out/snap384
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

INIT_SEED = 0
SPECS = {
    "a": dict(clip=dict(n_frames=16, H=384, W=224, seed=5), K=3, T=20, golden="loop_16f_384x224.npz"),
    "dense": dict(clip=dict(n_frames=8, H=384, W=224, seed=11, mask_keep=1.0, step=0.004, max_angle=0.012), K=3, T=10,
                  golden="loop_dense8f_384x224.npz"),
}
EVAL_STRIDE, EXPORT_STRIDE, CKPT_EPOCH_STRIDE, CKPT_STRIDE = 8, 2, 256, 16


def golden_path(spec):
    return os.path.join(REPO, "tests", "golden", SPECS[spec]["golden"])


def initial_state():
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    torch.manual_seed(INIT_SEED)
    return HourglassModel().state_dict()


def adam_snapshot(ft):
    """Weights, BatchNorm buffers, Adam moments and step count of a product fine-tuner, on the host."""
    torch.cuda.synchronize()
    opt = getattr(ft._step, "step", ft._step).opt      # GraphedFineTuneStep wraps the FineTuneStep that owns FlatAdam
    names = {id(p): n for n, p in ft.model.netG.named_parameters()}
    return {"state": {k: v.detach().cpu().clone() for k, v in ft.model.netG.state_dict().items()},
            "m1": {names[id(p)]: opt.exp_avg[o:o + p.numel()].detach().cpu().clone() for p, o in zip(opt._params, opt._offsets)},
            "m2": {names[id(p)]: opt.exp_avg_sq[o:o + p.numel()].detach().cpu().clone() for p, o in zip(opt._params, opt._offsets)},
            "k": int(opt.step_dev.item())}


def run_product(spec, work_dir, extra_args=()):
    """The product loop on the GPU: K + T epochs; returns (fine-tuner, snapshot at the end of epoch K, {epoch: plan}, clip path, range dir)."""
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.params import Video3dParamsParser
    S = SPECS[spec]
    K, T = S["K"], S["T"]
    path = os.path.join(work_dir, "clip")
    range_dir, pairs = msd.write_dataset(path, **S["clip"])
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", str(K + T), "--batch_size", "4", "--print_freq", "0", *extra_args])
    ft = DepthFineTuner(range_dir, list(range(S["clip"]["n_frames"])), params)
    ft.model.netG.load_state_dict(initial_state())
    snap, save = {}, ft.model.save

    def save_and_snapshot(file_name):      # end of every epoch (save_epoch_freq = 1), after its validation sweep
        save(file_name)
        if os.path.basename(file_name) == f"{K:04d}.pth":
            snap.update(adam_snapshot(ft))
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
    """EXACT, order-independent fingerprints of a snapshot: integer sums over the float32 BIT PATTERNS of every tensor (sum, sum of
    squares of the low 16 bits, xor-fold) -- the same numbers on any host, with any thread count.  (Rounds 4's fingerprints were fp64
    sums of the values: their last bits depend on the summation order, i.e. on the host's core count -- the same snapshot file failed
    its own check under a 2-core affinity mask.)"""
    out = {}
    for part in ("state", "m1", "m2"):
        s1 = s2 = x = 0
        for v in snap[part].values():
            if not v.is_floating_point():
                continue
            bits = v.detach().cpu().contiguous().float().view(torch.int32).to(torch.int64).reshape(-1)
            s1 += int(bits.sum())
            s2 += int(((bits & 0xffff) * (bits & 0xffff)).sum())
            x ^= int(bits.sum() * 2654435761 % (1 << 61)) ^ int(bits.numel())
        out[part] = np.array([s1, s2, x], np.int64)
    return out


def subsample(maps, stride):
    return np.ascontiguousarray(np.asarray(maps)[..., ::stride, ::stride])


def _ckpt_sample(file_name, stride):
    sd = torch.load(file_name, map_location="cpu")
    keys = [k for k in sd if sd[k].is_floating_point() and "uncertainty" not in k]
    sample = np.concatenate([sd[k].double().reshape(-1)[::stride].numpy() for k in keys])
    return sample, np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")], np.int64)


def collect(out_dir, n_pairs, K, T, skipped_pairs=0):
    """Artefacts of epochs K+1 .. K+T of a run (product or CPU loop), in the golden's (sub-sampled) form.  Epochs whose files are
    not there yet (an interrupted CPU run) end the collection: `res["epochs"]` is the list of complete ones."""
    from oracle import gen_golden_loop as G
    a = G.collect({"out_dir": out_dir, "steps": []})
    res, epochs = {}, []
    for e in range(K + 1, K + T + 1):
        tag = f"e{e:04d}_iter{e * n_pairs - skipped_pairs:06d}"
        ck = os.path.join(out_dir, "checkpoints", f"{e:04d}.pth")
        if f"val_{tag}_mean" not in a or not os.path.exists(ck):
            break
        for part in ("pairs", "reprojection", "disparity", "mean"):
            res[f"val_e{e}_{part}"] = a[f"val_{tag}_{part}"]
        res[f"evaldepth_e{e}"] = subsample(a[f"evaldepth_{tag}"], EVAL_STRIDE)
        res[f"ckpt_e{e}"], res["num_batches_tracked"] = _ckpt_sample(ck, CKPT_EPOCH_STRIDE)
        epochs.append(e)
    if epochs:
        res["ckpt_sample"], _ = _ckpt_sample(os.path.join(out_dir, "checkpoints", f"{epochs[-1]:04d}.pth"), CKPT_STRIDE)
    if "depth" in a and len(epochs) == T:
        res["depth"] = subsample(a["depth"], EXPORT_STRIDE)
    res["epochs"] = np.array(epochs, np.int64)
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


def stage_snapshot(spec, dst):
    S = SPECS[spec]
    K, T = S["K"], S["T"]
    os.makedirs(dst, exist_ok=True)
    work = tempfile.mkdtemp(prefix="cd384_")
    ft, snap, plans, _, _ = run_product(spec, work)
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
    flat["mask_coverage"] = np.array(float(ft.store.masks.float().mean().item()))
    for name, v in checksums(snap).items():
        flat["checksum_" + name] = v
    np.savez_compressed(os.path.join(dst, f"snap_{spec}.npz"), **flat)
    prod = collect(ft.out_dir, len(ft.store), K, T)
    np.savez_compressed(os.path.join(dst, f"product_{spec}.npz"), **prod)
    print(f"[{spec}] snapshot at step", snap["k"], "->", os.path.getsize(os.path.join(dst, f"snap_{spec}.npz")) / 1e6, "MB; mask coverage",
          float(flat["mask_coverage"]), {k: v.tolist() for k, v in checksums(snap).items()})
    for e in prod["epochs"]:
        print(f"[{spec}] product epoch {e} mean", prod[f"val_e{e}_mean"].tolist())


def _cpu_run(spec, src, dtype, device="cpu"):
    """oracle/cpu_loop.py continued from the snapshot in `src` for T epochs in `dtype`; returns (artefacts, snapshot checksums, extras).
    The run lives in <src>/cpu_<spec>_<dtype>; if that directory already holds a run (finished or interrupted) it is COLLECTED, not redone.
    device = "cuda": the SAME torch program evaluated by ATen on the GPU (its own fp64 kernels; the loss stays the C oracle on the host) --
    a 200-step fp64 continuation takes minutes there and ~7 hours on the build container's 8 cores; see stage_golden_gpu."""
    import make_synthetic_dataset as msd
    from consistent_depth_amd.loaders.video_dataset import VideoDataset, load_color
    from oracle import conv64, cpu_loop
    S = SPECS[spec]
    K, T, CLIP = S["K"], S["T"], S["clip"]
    conv64.ENABLED = device == "cpu"     # the dgemm formulation of the fp64 convolution: 5x faster on the build container's 8 cores
    z = np.load(os.path.join(src, f"snap_{spec}.npz"))
    tables = json.loads(str(z["tables"]))
    snap = {part: _unpack(z[part], tables[part]) for part in ("state", "m1", "m2")}
    snap["k"] = int(z["k"])
    for name, v in json.loads(str(z["ints"])).items():
        snap["state"][name] = torch.tensor(v, dtype=torch.int64)
    plans = {int(e): p for e, p in json.loads(str(z["plans"])).items()}
    cs = checksums(snap)
    for name, v in cs.items():
        if z["checksum_" + name].dtype == np.int64:          # (snapshots written before the exact fingerprints carry fp64 sums: not compared)
            assert np.array_equal(v, z["checksum_" + name]), name
    work = os.path.join(src, f"cpu_{spec}_{'f64' if dtype == torch.float64 else 'f32'}" + ("" if device == "cpu" else "_gpu"))
    path = os.path.join(work, "clip")
    out = os.path.join(work, "cpu")
    fresh = not os.path.exists(out)
    os.makedirs(work, exist_ok=True)
    range_dir, pairs = msd.write_dataset(path, **CLIP)
    ds = VideoDataset(path, os.path.join(range_dir, "metadata_scaled.npz"))
    assert len(ds) == len(pairs)
    step_losses = []
    if fresh:
        ds_idx = {tuple(int(v) for v in pr): i for i, pr in enumerate(ds.flow_indices)}
        plans = {e: [[ds_idx[tuple(pr)] for pr in batch] for batch in p] for e, p in plans.items()}
        lp = cpu_loop.CpuLoop(ds, snap["state"], out, dtype=dtype, device=device)
        lp.ft.set_adam_state(snap["m1"], snap["m2"], snap["k"])
        lp.total_iters = K * len(ds)
        lp.fine_tune(T, lambda e: plans[K + e], start_epoch=K)
        lp.save_depth(out, list(range(CLIP["n_frames"])), lambda f: load_color(ds.color_fmt.format(f)))
        step_losses = [l for _, _, l in lp.step_losses]
        np.save(os.path.join(work, "step_losses.npy"), np.array(step_losses, np.float64))
    elif os.path.exists(os.path.join(work, "step_losses.npy")):
        step_losses = np.load(os.path.join(work, "step_losses.npy")).tolist()
    res = collect(out, len(ds), K, T)
    extras = {"clip": np.array(json.dumps(CLIP)), "K": np.array(K), "T": np.array(len(res["epochs"])),
              "k_steps": np.array(snap["k"]), "plans": z["plans"], "eval_stride": np.array(EVAL_STRIDE), "export_stride": np.array(EXPORT_STRIDE),
              "ckpt_stride": np.array(CKPT_STRIDE), "ckpt_epoch_stride": np.array(CKPT_EPOCH_STRIDE),
              "mask_coverage": z["mask_coverage"],
              "pair_order": np.array([list(p) for p in ds.flow_indices], np.int64),
              "step_losses": np.array(step_losses, np.float64)}
    for k in list(res):
        if res[k].dtype == np.float64 and res[k].size > 4096:
            res[k] = res[k].astype(np.float32)       # maps and the checkpoint samples: the artefacts themselves are fp32 files
    return res, cs, extras


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300))


def distances(got, z, epochs, prefix=""):
    """Relative-L1 distance of every artefact of `got` to the golden `z`, per epoch: the rows of profiles/parity_20ep_r05.txt and the
    quantities tests/test_loop_gpu.py bounds.  prefix = "ref32_": against the artefacts of the reference's OWN arithmetic (the fp32 CPU
    continuation stored next to the fp64 ones since round 6) -- the comparison BASELINE.json literally names."""
    rows = {}
    for e in epochs:
        rows[int(e)] = {
            "mean": _rel(got[f"val_e{e}_mean"], z[f"{prefix}val_e{e}_mean"]),
            "perpair": max(_rel(got[f"val_e{e}_{p}"], z[f"{prefix}val_e{e}_{p}"]) for p in ("reprojection", "disparity")),
            "perpair_max": float(max(np.abs(np.asarray(got[f"val_e{e}_{p}"], np.float64) - z[f"{prefix}val_e{e}_{p}"]).max()
                                     / np.abs(z[f"{prefix}val_e{e}_{p}"]).mean() for p in ("reprojection", "disparity"))),
            "evaldepth": _rel(got[f"evaldepth_e{e}"], z[f"{prefix}evaldepth_e{e}"]),
            "ckpt": _rel(got[f"ckpt_e{e}"], z[f"{prefix}ckpt_e{e}"]),
        }
    return rows


def stage_golden(spec, src):
    res, cs, extras = _cpu_run(spec, src, torch.float64)
    res.update(extras)
    for name, v in cs.items():
        res["checksum_" + name] = v
    out = golden_path(spec)
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out) / 1e6, "MB; epochs", res["epochs"].tolist())
    prod = os.path.join(src, f"product_{spec}.npz")
    if os.path.exists(prod):
        p = np.load(prod)
        for e, row in distances(p, res, res["epochs"]).items():
            print(f"[{spec}] product (snapshot run) vs fp64, epoch {e}: " + "  ".join(f"{k} {v:.3e}" for k, v in row.items()))


def stage_golden_gpu(spec, src):
    """The fp64 ground truth computed by the oracle's torch program ON THE GPU (ATen's fp64 kernels -- im2col + dgemm convolutions --
    not a line of consistent_depth_amd; the loss is still the plain-C oracle on the host).  fp64 arithmetic differs between devices
    only in summation order (1e-13 per operation); `crosscheck` measures it against whatever the CPU run has finished and stores the
    distances in the golden.  (Why: 200 fp64 steps at 384x224 are ~7 hours on the build container's 8 cores.)"""
    res, cs, extras = _cpu_run(spec, src, torch.float64, device="cuda")
    res.update(extras)
    res["truth_device"] = np.array("cuda: oracle/cpu_loop.py's torch program on ATen fp64 kernels, C-oracle loss on the host")
    for name, v in cs.items():
        res["checksum_" + name] = v
    out = golden_path(spec)
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out) / 1e6, "MB; epochs", res["epochs"].tolist())


def stage_crosscheck(spec, src):
    """Distances of the golden (GPU fp64) to the CPU fp64 run of the same continuation, for every epoch the CPU run has completed
    (build container; the CPU run may be interrupted).  Stored in the golden as cpu64_crosscheck_{epochs, mean, perpair, evaldepth, ckpt}."""
    S = SPECS[spec]
    out = golden_path(spec)
    z = dict(np.load(out))
    work = os.path.join(src, f"cpu_{spec}_f64", "cpu")
    cpu = collect(work, len(z["pair_order"]), S["K"], S["T"])
    epochs = [int(e) for e in cpu["epochs"] if int(e) in set(z["epochs"].tolist())]
    rows = distances(cpu, z, epochs)
    for e, row in rows.items():
        print(f"[{spec}] CPU fp64 vs golden (GPU fp64), epoch {e}: " + "  ".join(f"{k} {v:.3e}" for k, v in row.items()))
    z["cpu64_crosscheck_epochs"] = np.array(epochs, np.int64)
    for name in ("mean", "perpair", "evaldepth", "ckpt"):
        z["cpu64_crosscheck_" + name] = np.array([rows[e][name] for e in epochs], np.float64)
    np.savez_compressed(out, **z)


def stage_ref32(spec, src):
    """The reference's OWN arithmetic: the same continuation in fp32 on the CPU (torch's CPU kernels, the path BASELINE.json names).
    Stored next to the fp64 golden: `ref32dist_<artefact>` = its distance to the fp64 run (the yardstick of rounds 4-5) and, since round
    6, the ARTEFACTS themselves as `ref32_<artefact>` (sampled like the fp64 ones), so that the product can be compared with the
    reference path DIRECTLY (tests/test_loop_gpu.py, third column; tools/parity_decompose.py)."""
    res, cs, _ = _cpu_run(spec, src, torch.float32)
    out = golden_path(spec)
    z = dict(np.load(out))
    for name, v in cs.items():
        assert np.array_equal(v, z["checksum_" + name]), name
    assert [int(e) for e in res["epochs"]] == [int(e) for e in z["epochs"]], (res["epochs"], z["epochs"])
    for k, v in res.items():
        if v.dtype.kind == "f" and k in z and z[k].shape == v.shape:
            z["ref32dist_" + k] = np.array(_rel(v, z[k]))
            z["ref32_" + k] = v.astype(np.float32) if v.size > 4096 else v
            print(f"[{spec}] reference fp32 vs fp64  {k:24s} {float(z['ref32dist_' + k]):.3e}")
    for e in z["epochs"]:      # the per-pair maximum, like distances()
        if f"val_e{e}_reprojection" in res:
            z[f"ref32dist_perpair_max_e{e}"] = np.array(distances(res, z, [e])[int(e)]["perpair_max"])
    np.savez_compressed(out, **z)


def stage_run32(spec, src):
    """Only the fp32 continuation itself (its work directory persists; `ref32` collects it once the fp64 golden exists)."""
    _cpu_run(spec, src, torch.float32)


if __name__ == "__main__":
    {"snapshot": stage_snapshot, "golden": stage_golden, "golden_gpu": stage_golden_gpu, "crosscheck": stage_crosscheck, "ref32": stage_ref32,
     "run32": stage_run32}[sys.argv[1]](sys.argv[2], sys.argv[3])
