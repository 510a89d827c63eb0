"""CPU: host-side logic added around the HIP path -- the launch-shape tuning cache and the graph step's input
bookkeeping.  No GPU, no compute calls."""
import json

import pytest
import torch


def test_tuning_cache_round_trips_through_a_file(tmp_path, monkeypatch):
    from consistent_depth_amd.ops import conv
    path = tmp_path / "tune.json"
    monkeypatch.setenv("CD_AMD_CONV_TUNE_CACHE", str(path))
    saved, loaded = dict(conv._TUNED), conv._TUNE_CACHE_LOADED[0]
    try:
        conv._TUNED.clear()
        key = (7, 32, 64, 8, 96, 56, 1, 1, 1, 0, 224, 224)
        conv._TUNED[key] = (8, 1)
        conv._TUNED[(1, 128, 208, 8, 384, 224, 0, 1, 1, 0, 128, 432)] = None      # "no valid shape" is cached too
        conv._save_tune_cache()
        assert sorted(map(str, json.load(open(path)).values())) == ["None", "[8, 1]"]
        conv._TUNED.clear()
        conv._TUNE_CACHE_LOADED[0] = False
        conv._load_tune_cache()
        assert conv._TUNED[key] == (8, 1)
        assert conv._TUNED[(1, 128, 208, 8, 384, 224, 0, 1, 1, 0, 128, 432)] is None
        # a corrupt file is ignored (shapes are simply measured again)
        path.write_text("{not json")
        conv._TUNED.clear()
        conv._TUNE_CACHE_LOADED[0] = False
        conv._load_tune_cache()
        assert conv._TUNED == {}
    finally:
        conv._TUNED.clear()
        conv._TUNED.update(saved)
        conv._TUNE_CACHE_LOADED[0] = loaded


def test_autotune_switch(monkeypatch):
    from consistent_depth_amd.ops import conv
    monkeypatch.setenv("CD_AMD_CONV_AUTOTUNE", "0")
    assert not conv.autotune_enabled()
    assert conv.tuned_config(3, 32, 64, 2, 24, 32, "cpu") is None      # disabled: library heuristic, nothing is launched
    monkeypatch.delenv("CD_AMD_CONV_AUTOTUNE")
    assert conv.autotune_enabled()


def _meta(B=2, H=8, W=12, with_cache=False):
    g = {"flows": [torch.zeros(B, 2, H, W), torch.ones(B, 2, H, W)], "masks": [torch.ones(B, 1, H, W), torch.ones(B, 1, H, W)]}
    if with_cache:
        g["mask_sums"] = torch.full((B, 2), float(H * W))
    return {"intrinsics": torch.zeros(B, 2, 4), "extrinsics": torch.zeros(B, 2, 3, 4), "geometry_consistency": g}


def test_graph_step_input_bookkeeping():
    from consistent_depth_amd import engine
    m = _meta()
    flat = engine._flatten(m)
    assert [p for p, _ in flat] == [("extrinsics",), ("geometry_consistency", "flows", 0), ("geometry_consistency", "flows", 1),
                                   ("geometry_consistency", "masks", 0), ("geometry_consistency", "masks", 1), ("intrinsics",)]
    clone = engine._clone_tree(m)
    for (pa, a), (pb, b) in zip(flat, engine._flatten(clone)):
        assert pa == pb and torch.equal(a, b) and a.data_ptr() != b.data_ptr()
    sig = engine.GraphedFineTuneStep._signature
    img = torch.zeros(2, 2, 3, 8, 12)
    assert sig(img, m) == sig(img.clone(), _meta())
    assert sig(img, m) != sig(img, _meta(with_cache=True))            # optional cached constants change the graph
    assert sig(img, m) != sig(torch.zeros(3, 2, 3, 8, 12), _meta(B=3))  # so does the batch size


def test_graph_step_runs_eagerly_first_and_falls_back_when_capture_is_unavailable():
    """Without a device there is nothing to capture: the wrapper must hand the first calls to the eager step and, when
    capture raises, stay eager for good (same kernels) instead of failing."""
    from consistent_depth_amd import engine

    class _Step:
        world = 1
        calls = 0

        def __call__(self, images, metadata):
            self.calls += 1
            return torch.tensor(float(self.calls)), {}

        def _grads(self, images, metadata):
            raise RuntimeError("no device")

    st = _Step()
    g = engine.GraphedFineTuneStep(st, eager_steps=2)
    img, m = torch.zeros(1, 2, 3, 8, 12), _meta(B=1)
    g._capture = lambda images, metadata: (_ for _ in ()).throw(RuntimeError("stream capture unsupported"))
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        for expected in (1.0, 2.0, 3.0, 4.0):
            loss, _ = g(img, m)
            assert loss.item() == expected
    finally:
        torch.cuda.synchronize = real_sync
    assert g.graphed is False and "capture unsupported" in g.capture_error


def test_every_python_file_compiles():
    """tools/, oracle/, bench.py, main.py ... are only partly imported by the other tests: byte-compile all of them."""
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 0
    for base, dirs, files in os.walk(root):
        dirs[:] = [d for d in dirs if d not in (".git", "__pycache__", "gpurun_out", "build")]
        for f in files:
            if f.endswith(".py"):
                py_compile.compile(os.path.join(base, f), doraise=True)
                n += 1
    assert n > 40


def test_bench_cli_contract():
    """bench.py must accept exactly the driver's flags and default to N=1 (the run itself needs the GPU)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_cli", os.path.join(root, "bench.py"))
    src = open(os.path.join(root, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    assert spec is not None and '"cpu_baseline"' in src and '"roofline"' in src and "higher_is_better" in src
    del sys


def test_bench_flop_count_matches_the_convolutions_a_forward_pass_executes(monkeypatch):
    """`roofline_conv` in the bench line prices bench.mc_conv_macs(H, W) multiply-adds per image.  Counted here from the
    convolutions an actual forward pass of the restated hourglass executes (every F.conv2d call of oracle/hourglass_ref.forward,
    at a small size; the count is proportional to H W): the table walk must give the same number, minus the confidence head the
    fine-tuning path never runs, and 52.78 GMAC at the headline 384 x 224."""
    import importlib.util
    import os
    import torch.nn.functional as F
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    from oracle import hourglass_ref as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    H, W = 64, 32
    macs = [0]
    real = F.conv2d

    def counting(x, w, *a, **k):
        y = real(x, w, *a, **k)
        macs[0] += y.shape[0] * y.shape[2] * y.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        return y
    monkeypatch.setattr(R.F, "conv2d", counting)
    torch.manual_seed(0)
    state = {k: v.double() for k, v in HourglassModel().state_dict().items()}
    R.forward(state, torch.rand(1, 3, H, W, dtype=torch.float64), training=True)
    head = H * W * 64 * 9            # the confidence head: run by the oracle's forward, not by the fine-tuning path
    assert macs[0] == R.conv_macs(H, W)
    assert bench.mc_conv_macs(H, W) == macs[0] - head
    assert abs(bench.mc_conv_macs(384, 224) / 1e9 - 52.783) < 1e-3


def test_a_captured_collective_does_not_fall_back(monkeypatch):
    """CD_AMD_DP_GRAPH_COLLECTIVE=1 puts the gradient all-reduce inside the step graph.  A rank whose capture fails must not
    quietly run eager steps while the others replay a graph with the collective inside (their call sequences would diverge).  Round 5
    raised; round 6: the ranks AGREE after the capture attempt (parallel.all_agree) and drop to the eager exchange TOGETHER -- here, with no
    process group, the one rank's own verdict decides (the two-rank case: tests/test_parallel_cpu.py).  Without the opt-in (or with one
    rank) the plain eager fallback stays."""
    from consistent_depth_amd import engine

    class _Step:
        def __init__(self, world):
            self.world, self.calls, self.updates = world, 0, 0

        def __call__(self, images, metadata):
            self.calls += 1
            return torch.tensor(float(self.calls)), {}

        def _update(self, guard):
            self.updates += 1

    def broken_capture(images, metadata):
        raise RuntimeError("stream capture unsupported")

    img, m = torch.zeros(1, 2, 3, 8, 12), _meta(B=1)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setenv("CD_AMD_DP_GRAPH_COLLECTIVE", "1")
    g = engine.GraphedFineTuneStep(_Step(world=2), eager_steps=1)
    assert g.graph_collective is True
    g._capture = broken_capture
    g(img, m)                                  # the eager step
    out = g(img, m)                            # capture fails -> consensus "not everybody has a graph" -> eager exchange, and (capture still broken) eager steps
    assert out[0].item() == 2.0 and g.graph_collective is False and g.graphed is False
    assert "eager exchange on every rank" in g.capture_error or "stream capture unsupported" in g.capture_error
    one = engine.GraphedFineTuneStep(_Step(world=1), eager_steps=1)
    assert one.graph_collective is False       # nothing to put inside with one rank
    monkeypatch.setenv("CD_AMD_DP_GRAPH_COLLECTIVE", "0")
    st = _Step(world=2)
    g = engine.GraphedFineTuneStep(st, eager_steps=1)
    assert g.graph_collective is False
    g._capture = broken_capture
    for expected in (1.0, 2.0, 3.0):
        assert g(img, m)[0].item() == expected
    assert g.graphed is False and st.calls == 3


def test_engine_reference_golden_matches_the_network_it_is_for():
    """tests/golden/engine_ref_8x384x224.npz (tests/bg_reference.py --golden): one sampled fp32 array per parameter gradient of the
    hourglass, the sampled prediction, the full BatchNorm running statistics -- keyed and sized for the network
    tests/test_hourglass_engine_gpu.py::test_engine_matches_autograd builds, with the sampling rule the test applies to the engine's tensors."""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bg_reference as R
    from consistent_depth_amd.monodepth.hourglass import HourglassModel
    g = R.load_golden()
    assert g is not None and int(g["sampled"]) == R.SAMPLE
    net = HourglassModel()
    n_grad = 0
    for name, p in net.named_parameters():
        key = "grad:" + name
        if name.startswith("uncertainty_layer"):
            assert key not in g                                   # the unused head receives no gradient
            continue
        assert key in g, name
        want = R.sample(torch.zeros(p.numel())).shape[0]
        assert g[key].shape == (want,) and g[key].dtype == np.float32 and np.isfinite(g[key]).all(), name
        assert want >= min(p.numel(), R.SAMPLE)
        n_grad += 1
    assert n_grad == len([k for k in g if k.startswith("grad:")]) > 300
    assert g["pred"].shape == (R.sample(torch.zeros(R.N * R.H * R.W)).shape[0],) and np.abs(g["pred"]).max() > 0
    stats = [k for k in g if k.startswith("stat:")]
    sd = net.state_dict()
    assert len(stats) == 2 * 155
    for k in stats:
        assert g[k].shape == tuple(sd[k[len("stat:"):]].shape)
    # numpy and torch sample alike
    a = np.arange(100000, dtype=np.float32)
    assert np.array_equal(R.sample(a), R.sample(torch.as_tensor(a)).numpy())


def test_wgrad_table_orders_heaviest_first_and_accumulates_block_ranges(monkeypatch):
    """ops/conv.py::WgradTable.build(): descriptors grouped by kernel class, inside a class ordered by the work of ONE workgroup
    (image tiles it walks x taps) descending -- workgroups are dispatched in table order, the long ones must not start last --, block_end =
    running sum of the members' workgroup counts, at most 64 descriptors per launch.  (The descriptors themselves come from the C library:
    injected here.)"""
    import numpy as np
    import torch
    from consistent_depth_amd.ops import conv
    t = conv.WgradTable(torch.device("cpu"))
    rng = np.random.default_rng(0)

    def desc(klass, ks, N, tiles_x, tiles_y, splits, blocks):
        d = np.zeros(1, np.dtype(conv.WgradTable._DT))
        d[0]["klass"], d[0]["ks"], d[0]["N"], d[0]["tiles_x"], d[0]["tiles_y"], d[0]["splits"], d[0]["blocks"] = klass, ks, N, tiles_x, tiles_y, splits, blocks
        return d
    members = [desc(3, 7, 8, 7, 64, 64, 512), desc(3, 7, 8, 1, 3, 24, 192), desc(3, 7, 8, 4, 19, 64, 512), desc(0, 3, 8, 7, 64, 128, 512)]
    members += [desc(2, 5, 8, 2, 5, int(rng.integers(8, 64)), int(rng.integers(64, 512))) for _ in range(70)]
    t._descs = [members[i] for i in rng.permutation(len(members))]
    launches = []
    monkeypatch.setattr(torch, "from_numpy", lambda a: _Keep(a, launches))
    t.build()
    by_class = {}
    for klass, tab, n, total in t._launches:
        by_class.setdefault(klass, []).append((tab.array, n, total))
    assert sorted(by_class) == [0, 2, 3]
    assert [n for _, n, _ in by_class[2]] == [64, 6]                      # 70 descriptors of class 2: two launches
    for klass, parts in by_class.items():
        for arr, n, total in parts:
            rec = arr.view(np.dtype(conv.WgradTable._DT))
            assert len(rec) == n and list(rec["block_end"]) == list(np.cumsum(rec["blocks"])) and total == int(rec["block_end"][-1])
            assert set(rec["klass"]) == {klass}
            work = [-(-(int(r["N"]) * int(r["tiles_x"]) * int(r["tiles_y"])) // int(r["splits"])) * int(r["ks"]) ** 2 for r in rec]
            if len(parts) == 1:
                assert work == sorted(work, reverse=True)


class _Keep:
    """Stand-in for the uploaded table: keeps the host bytes (`.to()` returns itself)."""

    def __init__(self, array, log):
        self.array = array
        log.append(self)

    def to(self, *_a, **_k):
        return self


def test_merged_dispatch_is_off_without_autotune(monkeypatch):
    """ops/conv.py::tuned_multi takes a decision only by timing; with the autotuner disabled (or CD_AMD_CONV_MULTI=0) it returns None
    without touching the device: the engine then launches the branches one by one."""
    from consistent_depth_amd.ops import conv
    monkeypatch.setenv("CD_AMD_CONV_AUTOTUNE", "0")
    assert conv.tuned_multi([], []) is None
    monkeypatch.setenv("CD_AMD_CONV_AUTOTUNE", "1")
    monkeypatch.setenv("CD_AMD_CONV_MULTI", "0")
    assert conv.tuned_multi([], []) is None


def test_gpu_suite_runs_the_live_cpu_references_last_and_within_the_budget(monkeypatch):
    """tests/conftest.py: the three tests whose cost is a live CPU reference on the GPU box's host go to the end of the session, cheapest
    first, and are skipped -- with the numbers in the reason -- only when the time left cannot hold their estimate."""
    import conftest

    class Item:
        def __init__(self, nodeid):
            self.nodeid = nodeid
    heavy = ["tests/" + k for k in conftest._HEAVY]
    ids = [heavy[2], "tests/test_a_gpu.py::test_x", heavy[0], "tests/test_b_gpu.py::test_y[hip]", heavy[1], "tests/test_c_gpu.py::test_z"]
    out = [it.nodeid for it in conftest.order_heavy_last([Item(i) for i in ids])]
    assert out[:3] == [ids[1], ids[3], ids[5]]                                   # the others keep their order
    assert out[3:] == sorted(heavy, key=conftest._heavy_cost)                    # cheapest first
    # a parametrisation that is not in the table is not heavy
    assert conftest._heavy_cost("tests/test_finetune_gpu.py::test_short_finetune_matches_cpu_reference[torch]") == 0
    cost = conftest._heavy_cost(heavy[2])
    assert conftest.budget_verdict(heavy[2], 1050 - cost, 1050) is None
    why = conftest.budget_verdict(heavy[2], 1050 - cost + 1, 1050)
    assert why and "time budget" in why and str(cost) in why
    assert conftest.budget_verdict(heavy[2], 5000, 0) is None                    # CD_AMD_TEST_BUDGET_S=0: no guard
    assert conftest.budget_verdict("tests/test_a_gpu.py::test_x", 5000, 1050) is None
    assert sum(conftest._HEAVY.values()) + 330 < 1200                            # the estimates plus the rest of the suite (~290 s) fit the driver's 1200 s
    monkeypatch.setenv("CD_AMD_TEST_BUDGET_S", "0")
    assert conftest.budget_left() == float("inf")
    monkeypatch.setenv("CD_AMD_TEST_BUDGET_S", "100000")
    assert 0 < conftest.budget_left() < 100000                                    # counts from the start of the interpreter


def test_mc_checkpoint_layout_compatibility(tmp_path):
    """`checkpoints/mc.pth` as the reference's adapter consumes it (/root/reference/monodepth/mannequin_challenge_model.py:34-41: the file is
    handed to Pix2PixModel.load_network as-is; :71-73 saves netG.state_dict()): the key list and shapes of this repo's parameter container,
    of the oracle's independent restatement (oracle/hourglass_ref.py: the keys oracle/ref_loop.py serves to the reference adapter as the
    "downloaded" checkpoint) and of a round trip through both prefix conventions -- upstream wraps netG in nn.DataParallel, so real files
    carry `module.` on every key; files written by `save` of a single-device model do not.  (The upstream file itself is not available
    offline: what is pinned here is that every layout the adapter can meet loads, and that anything else is refused, not half-loaded.)"""
    import torch
    from consistent_depth_amd.monodepth.hourglass import HourglassModel, load_state_dict_any_prefix
    from oracle import hourglass_ref
    torch.manual_seed(3)
    net = HourglassModel(3)
    sd = net.state_dict()
    # ---- the container's layout = the architecture of SURVEY.md A.3: 157 convolutions, 155 BatchNorms, 5 357 730 parameters
    convs = [k for k in sd if k.endswith(".weight") and sd[k].dim() == 4]
    assert len(convs) == 157 and sum(p.numel() for p in net.parameters()) == 5357730
    assert sum(1 for k in sd if k.endswith("running_mean")) == 155 and sum(1 for k in sd if k.endswith("num_batches_tracked")) == 155
    assert {"seq.0.weight", "seq.0.bias", "seq.1.weight", "seq.1.bias", "seq.1.running_mean", "pred_layer.weight", "pred_layer.bias",
            "uncertainty_layer.0.weight", "uncertainty_layer.0.bias"} <= set(sd)
    assert tuple(sd["seq.0.weight"].shape) == (128, 3, 7, 7) and tuple(sd["pred_layer.weight"].shape) == (1, 64, 3, 3)
    # ---- every key the oracle's functional restatement reads exists with the shape it needs: a forward through it is the check
    x = torch.rand(1, 3, 32, 32, dtype=torch.float64)
    pred, _ = hourglass_ref.forward({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, x, training=False)
    assert tuple(pred.shape) == (1, 1, 32, 32) and torch.isfinite(pred).all()
    # ---- both prefix conventions load, bit for bit
    for prefix in ("", "module."):
        fn = str(tmp_path / f"mc_{len(prefix)}.pth")
        torch.save({prefix + k: v for k, v in sd.items()}, fn)
        other = HourglassModel(3)
        load_state_dict_any_prefix(other, torch.load(fn, map_location="cpu"))
        for k, v in other.state_dict().items():
            assert torch.equal(v, sd[k]), (prefix, k)
    # ---- anything else is refused: a missing key, an unexpected key, a wrong shape
    bad = dict(sd); bad.pop("pred_layer.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        load_state_dict_any_prefix(HourglassModel(3), bad)
    bad = dict(sd); bad["seq.9.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        load_state_dict_any_prefix(HourglassModel(3), bad)
    bad = dict(sd); bad["seq.0.weight"] = torch.zeros(128, 3, 5, 5)
    with pytest.raises(RuntimeError, match="size mismatch"):
        load_state_dict_any_prefix(HourglassModel(3), bad)
    # ---- and the adapter's own save() writes the un-prefixed layout the reference's adapter writes (:71-73)
    assert list(torch.load(str(tmp_path / "mc_0.pth"), map_location="cpu")) == list(sd)
