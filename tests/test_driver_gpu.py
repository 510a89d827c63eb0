"""GPU, end to end: a synthetic clip written in the reference's on-disk layout is fine-tuned through the
same entry points the reference's pipeline calls (process.py:57,88,93): DepthFineTuner(range_dir, frames,
params).fine_tune() and .save_depth(), with the CLI parser producing `params`.  Checks the output contract
(files, formats, names) and that the loss goes down."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_cli_finetune_on_disk_dataset(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import make_tag
    from consistent_depth_amd.params import Video3dParamsParser
    from consistent_depth_amd.process import DatasetProcessor
    from consistent_depth_amd.utils import image_io

    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=6, H=64, W=48, seed=3)
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", "2", "--batch_size", "4", "--print_freq", "2"])
    assert make_tag(params) == "B0.1_R1.0_PL1-0_LR0.0004_BS4_Oadam"
    _, out_dir, frames = DatasetProcessor().process(params)
    assert out_dir == os.path.join(range_dir, "B0.1_R1.0_PL1-0_LR0.0004_BS4_Oadam")
    assert frames == list(range(6))

    # checkpoints: netG.state_dict() per epoch, loadable into the hourglass container
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    for e in (1, 2):
        sd = torch.load(os.path.join(out_dir, "checkpoints", f"{e:04d}.pth"), map_location="cpu")
        HourglassModel().load_state_dict(sd)
    # validation dumps before epoch 0 and after each epoch: loss json + inverse-depth raws
    n_iter = len(pairs) * 2
    losses = []
    for epoch, it in ((0, 0), (1, len(pairs)), (2, n_iter)):
        fn = os.path.join(out_dir, "eval", f"loss_e{epoch:04d}_iter{it:06d}.json")
        with open(fn) as f:
            d = json.load(f)
        assert set(d) == {"reprojection", "disparity", "mean"}
        assert set(d["reprojection"]) == {str([i, j]) for i, j in pairs}
        losses.append(d["mean"]["reprojection"] + d["mean"]["disparity"])
        for fr in range(6):
            inv = image_io.load_raw_float32_image(os.path.join(out_dir, "eval", f"depth_{fr:06d}_e{epoch:04d}_iter{it:06d}.raw"))
            assert inv.shape == (64, 48) and np.isfinite(inv).all() and (inv > 0).all()
    assert losses[2] < losses[0], losses   # test-time training reduces the geometric inconsistency
    # final depth export: inverse depth per frame
    for fr in range(6):
        inv = image_io.load_raw_float32_image(os.path.join(out_dir, "depth", f"frame_{fr:06d}.raw"))
        assert inv.shape == (64, 48) and np.isfinite(inv).all() and (inv > 0).all()


def test_pair_store_batches_match_video_dataset(tmp_path):
    """The HBM-resident store hands the loop exactly what the reference's VideoDataset + collate would."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    import torch
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.loaders.video_dataset import VideoDataset
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=5, H=32, W=48, seed=4)
    meta_file = os.path.join(range_dir, "metadata_scaled.npz")
    store = PairStore.from_directory(path, meta_file)
    ds = VideoDataset(path, meta_file)
    assert len(store) == len(ds) == len(pairs)
    by_pair = {tuple(p): i for i, p in enumerate(store.pair_indices())}
    for k in range(len(ds)):
        images, meta = ds[k]
        pid = by_pair[tuple(meta["geometry_consistency"]["indices"].tolist())]
        bi, bm = store.batch([pid])
        assert torch.equal(bi[0].cpu(), images)
        assert torch.equal(bm["intrinsics"][0].cpu(), meta["intrinsics"]) and torch.equal(bm["extrinsics"][0].cpu(), meta["extrinsics"])
        for d in range(2):
            assert torch.equal(bm["geometry_consistency"]["flows"][d][0].cpu(), meta["geometry_consistency"]["flows"][d])
            assert torch.equal(bm["geometry_consistency"]["masks"][d][0].cpu(), meta["geometry_consistency"]["masks"][d])
        assert bm["geometry_consistency"]["mask_sums"].shape == (1, 2)


def test_gather_into_graph_buffers_and_step_from_store():
    """cd_gather_pairs: a multi-pair batch equals row-by-row indexing of the store (bitwise, u8 masks widened to {0,1}), and a
    graph-replayed step fed by gather_into (straight into the graph's static inputs) trains like the batch()-fed eager step."""
    import argparse
    import torch
    from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    store = PairStore.synthetic(6, 64, 48, seed=2)
    ids = torch.tensor([3, 0, len(store) - 1, 1], device=store.device)
    images, meta = store.batch(ids)
    geom = meta["geometry_consistency"]
    pf = store.pair_frames[ids]
    assert torch.equal(images, store.color[pf])
    for d in range(2):
        assert torch.equal(geom["flows"][d], store.flows[ids][:, d])
        assert torch.equal(geom["masks"][d], store.masks[ids][:, d].float())
    assert torch.equal(meta["intrinsics"], store.intrinsics[pf]) and torch.equal(meta["extrinsics"], store.extrinsics[pf])
    assert torch.equal(geom["mask_sums"], store.mask_sums[ids]) and torch.equal(geom["tile_windows"], store.tile_windows[ids])
    assert geom["indices"].tolist() == [store.pair_indices()[i] for i in ids.tolist()]
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4, optimizer="Adam")
    model = get_depth_model("mc")(seed=0)
    model.train()
    step = GraphedFineTuneStep(FineTuneStep(model, params, world=1), eager_steps=1)
    w0 = step.step.opt.flat_param.clone()
    losses = []
    for it in range(5):
        pid = torch.tensor([(2 * it) % len(store), (2 * it + 1) % len(store)], device=store.device)
        loss, parts, md = step.step_from_store(store, pid)
        losses.append(loss.item())
        assert md["geometry_consistency"]["indices"].tolist() == [store.pair_indices()[i] for i in pid.tolist()]
    assert step.graphed is True, step.capture_error
    assert all(l == l for l in losses) and not torch.equal(w0, step.step.opt.flat_param)


def test_validation_sweep_values_match_cpu_reference(tmp_path):
    """eval_and_save (depth_fine_tuning.py:312-406 of the reference): the VALUES of eval/loss*.json (per pair and mean) and of
    the inverse-depth .raw files against the reference step restated on the CPU in fp64 (oracle/cpu_step pieces) on the same
    weights and the same batches (train-mode BatchNorm under no_grad, first sighting of a frame wins), written through the
    asynchronous writer."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    import torch
    from consistent_depth_amd import parallel
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.engine import FineTuneStep
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.params import Video3dParamsParser
    from consistent_depth_amd.utils import image_io
    from oracle import hourglass_ref, oracle
    oracle.build()
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=5, H=64, W=48, seed=6)
    params = Video3dParamsParser().parse(["--path", path, "--batch_size", "3"])
    ft = DepthFineTuner(range_dir, list(range(5)), params)
    ft.store = PairStore.from_directory(path, os.path.join(range_dir, "metadata_scaled.npz"))
    os.makedirs(os.path.join(ft.out_dir, "eval"), exist_ok=True)
    state = {k: v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu() for k, v in ft.model.netG.state_dict().items()}
    ft.model.train()
    step = FineTuneStep(ft.model, params, world=1)
    ft.eval_and_save(step, "_t")
    with open(os.path.join(ft.out_dir, "eval", "loss_t.json")) as f:
        got = json.load(f)
    want, first = {"reprojection": {}, "disparity": {}}, {}
    for ids in parallel.eval_shard(len(ft.store), 0, 1, 3):
        images, meta = ft.store.batch(ids)
        x = images.cpu().double().reshape(-1, 3, 64, 48)
        with torch.no_grad():
            pred, _ = hourglass_ref.forward(state, x, training=True, update_running_stats=False)
        depth = torch.exp(pred).reshape(len(ids), 2, 64, 48).numpy()
        g = meta["geometry_consistency"]
        out = oracle.consistency_loss(depth, [f.cpu().numpy() for f in g["flows"]], [m.cpu().numpy() for m in g["masks"]],
                                      meta["intrinsics"].cpu().numpy(), meta["extrinsics"].cpu().numpy(), 1.0, 0.1, want_grad=False)
        for b, pair in enumerate(g["indices"].tolist()):
            want["reprojection"][str(pair)] = out["reprojection"][b]
            want["disparity"][str(pair)] = out["disparity"][b]
            for k, fr in enumerate(pair):
                first.setdefault(fr, 1.0 / depth[b, k])
    assert set(got["reprojection"]) == set(want["reprojection"]) == {str(list(p)) for p in pairs}
    for name in ("reprojection", "disparity"):
        for key, v in want[name].items():
            assert got[name][key] == pytest.approx(v, rel=2e-3), (name, key)     # fp32 CNN forward at random init: ~2e-4 in depth
        assert got["mean"][name] == pytest.approx(np.mean(list(want[name].values())), rel=1e-3)
    for fr, inv in first.items():
        raw = image_io.load_raw_float32_image(os.path.join(ft.out_dir, "eval", f"depth_{fr:06d}_t.raw"))
        assert np.abs(raw - inv).sum() / np.abs(inv).sum() < 1e-3
    # save_depth: eval-mode BatchNorm, batched forward, same files as one frame at a time
    ft.save_depth(ft.out_dir, list(range(5)))
    ft.model.eval()
    with torch.no_grad():
        for fr in range(5):
            ref = 1.0 / ft.model.forward(ft.store.color[fr][None]).float().cpu().numpy().squeeze()
            raw = image_io.load_raw_float32_image(os.path.join(ft.out_dir, "depth", f"frame_{fr:06d}.raw"))
            np.testing.assert_allclose(raw, ref, rtol=1e-5)


def test_scene_scale_is_the_reference_scale_calibration():
    """PairStore.scale_scene_(s) multiplies the camera translations (scale_calibration.py:305-313 divides them by the calibrated
    scale); flows, masks and intrinsics do not depend on the scene's metric scale.  A size-independent property of the loss pins it:
    with depth and scene scaled together every reprojected pixel stays where it was -- the reprojection term is unchanged -- and
    every disparity is divided by s.  Checked at the headline shape on the fused HIP loss."""
    import torch
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.loss.consistency_loss import ConsistencyLoss

    class Opt:
        lambda_reprojection, lambda_view_baseline = 1.0, 1.0
    store = PairStore.synthetic(6, 384, 224, seed=5, device="cuda")
    ids = torch.arange(4, device="cuda")
    _, meta = store.batch(ids)
    g = torch.Generator(device="cuda").manual_seed(0)
    depth = torch.as_tensor(store.gt_depth, dtype=torch.float32, device="cuda")[store.pair_frames[ids]] * \
        (1.0 + 0.05 * torch.randn(4, 2, 384, 224, device="cuda", generator=g))
    crit = ConsistencyLoss(Opt())
    _, p0 = crit(depth, meta)
    s = 0.0123
    ext0 = store.extrinsics.clone()
    store.scale_scene_(s)
    assert torch.equal(store.extrinsics[..., :3], ext0[..., :3]) and torch.allclose(store.extrinsics[..., 3], ext0[..., 3] * s, rtol=1e-7, atol=0)
    _, meta_s = store.batch(ids)
    _, p1 = crit(depth * s, meta_s)
    np.testing.assert_allclose(p1["reprojection"].cpu().numpy(), p0["reprojection"].cpu().numpy(), rtol=2e-4)
    np.testing.assert_allclose(p1["disparity"].cpu().numpy() * s, p0["disparity"].cpu().numpy(), rtol=2e-4)


class _ScalarLog:
    def __init__(self):
        self.points = []

    def add_scalar(self, tag, value, n):
        self.points.append((tag, int(n)))


def test_training_scalars_are_logged_at_the_running_pair_count(tmp_path):
    """The reference advances total_iters by the batch BEFORE it logs the step (depth_fine_tuning.py:285-288): every step of an
    epoch lands on its own global position, in pairs.  (Round 3 logged a whole epoch at the epoch-start value.)"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.params import Video3dParamsParser
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=6, H=64, W=48, seed=3)
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", "2", "--batch_size", "4", "--print_freq", "1"])
    log = _ScalarLog()
    DepthFineTuner(range_dir, list(range(6)), params).fine_tune(writer=log)
    got = [n for tag, n in log.points if tag == "Train/loss"]
    n, want, pos = len(pairs), [], 0
    for _epoch in range(2):
        for s in range(0, n, 4):
            pos += min(4, n - s)
            want.append(pos)
    assert got == want, (got, want)


def test_training_scalars_skip_nan_steps_between_prints(tmp_path):
    """A NaN loss skips the step AND the `total_iters` advance in the reference (depth_fine_tuning.py:278-285): the training points after it
    keep their exact positions also when the NaN step was not a print step (its loss is only read at the end of the epoch) -- the scalars
    of an epoch are written once its NaN mask is known (round 4 corrected the running position on print steps only)."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.loaders.pair_store import PairStore
    from consistent_depth_amd.params import Video3dParamsParser
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=8, H=64, W=48, seed=3)
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", "2", "--batch_size", "2", "--print_freq", "2"])
    store = PairStore.from_directory(path, os.path.join(range_dir, "metadata_scaled.npz"))
    poisoned = 5
    store.flows[poisoned, 0, 0, 3, 3] = float("nan")          # every batch holding this pair has a NaN loss
    ft = DepthFineTuner(range_dir, list(range(8)), params, store=store)
    log = _ScalarLog()
    ft.fine_tune(writer=log)
    got = [n for tag, n in log.points if tag == "Train/loss"]
    want, pos = [], 0
    for epoch in range(2):
        for it, ids in enumerate(ft.epoch_plan(epoch)):
            bad = poisoned in ids
            if not bad:
                pos += len(ids)
            if it % 2 == 0 and not bad:
                want.append(pos)
    assert got == want and len(got) > 4, (got, want)
    # the validation files carry the same counter
    assert os.path.exists(os.path.join(ft.out_dir, "eval", f"loss_e0002_iter{pos:06d}.json"))


def test_parameter_only_objective_validates_without_per_pair_entries(tmp_path):
    """lambda_reprojection = lambda_view_baseline = 0, lambda_parameter > 0: JointLoss has no ConsistencyLoss term
    (joint_loss.py:20-24), so the validation sweep has no per-pair entries to index (round 3 raised KeyError)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import make_synthetic_dataset as msd
    from consistent_depth_amd.depth_fine_tuning import DepthFineTuner
    from consistent_depth_amd.params import Video3dParamsParser
    path = str(tmp_path / "clip")
    range_dir, pairs = msd.write_dataset(path, n_frames=4, H=64, W=48, seed=5)
    params = Video3dParamsParser().parse(["--path", path, "--num_epochs", "1", "--batch_size", "4", "--lambda_reprojection", "0",
                                          "--lambda_view_baseline", "0", "--lambda_parameter", "1.0"])
    ft = DepthFineTuner(range_dir, list(range(4)), params)
    ft.fine_tune()
    with open(os.path.join(ft.out_dir, "eval", f"loss_e0001_iter{len(pairs):06d}.json")) as f:
        assert json.load(f) == {"mean": {}}
    assert os.path.exists(os.path.join(ft.out_dir, "eval", f"depth_000000_e0001_iter{len(pairs):06d}.raw"))
