"""CPU checks of the row-sweep loss kernel (consistent_depth_amd/csrc/loss_sweep_core.h) without a GPU:

* the per-pair PLAN (schedule of row groups and ring windows): invariants the kernel relies on, for consistent, unrelated,
  wild and degenerate flows;
* the phase functions themselves, executed sequentially on the host by tests/emul (the same header hipcc compiles into the
  kernel): loss and gradient against the reference-run goldens and the fp64 oracle, through the fast path, the forced
  slow path, tiny rings (overflow list) and every pixels-per-thread setting.
"""
import numpy as np
import pytest

from conftest import golden_loss_cases, load_loss_case
from emul import build as E


def _check_plan(g, items, lo, hi):
    G, R, H, NG, SMAX = g["G"], g["R"], g["H"], g["NG"], g["SMAX"]
    assert len(items) <= g["max_items"]
    done = [[], []]
    w_prev = [0, 0]
    outside = 0
    for p0, p1, w0, w1 in items.tolist():
        w, p = [w0, w1], [p0, p1]
        for f in range(2):
            assert w[f] >= w_prev[f] and w[f] - w_prev[f] <= SMAX, "windows slide forward by at most SMAX rows per item"
            if p[f] >= 0:
                assert p[f] % G == 0
                # rows entering during an item are not used by it: the usable part of a ring ends at the PREVIOUS window's top
                assert w[f] <= p[f] and p[f] + G <= w_prev[f] + R, "a group's own rows live in its frame's ring"
                gi = p[f] // G
                done[f].append(gi)
                if hi[f, gi] >= 0 and not (lo[f, gi] >= w[1 - f] and hi[f, gi] < w_prev[1 - f] + R):
                    outside += 1
        w_prev = w
    for f in range(2):
        assert done[f] == list(range(NG)), "every row group exactly once, in order"
        assert w_prev[f] >= H, "all gradient rows have left the rings"
    return outside


@pytest.mark.parametrize("gen,kw,max_outside", [("scene", {}, 0), ("scene", {"frame_gap": 12}, 0), ("pair", {}, None),
                                                ("pair", {"noise_px": 40.0}, None)])
@pytest.mark.parametrize("H,W,pxt", [(384, 224, 2), (384, 224, 4), (384, 224, 1), (224, 384, 2), (96, 128, 2)])
def test_plan_invariants(gen, kw, max_outside, H, W, pxt):
    from consistent_depth_amd import synthetic
    g = E.geo(H, W, pxt)
    assert g is not None
    batch = (synthetic.make_scene_batch if gen == "scene" else synthetic.make_pair_batch)(3, H, W, seed=5, **kw)
    for b in range(3):
        items, lo, hi = E.plan(g, batch["flows"][0][b], batch["flows"][1][b], batch["masks"][0][b], batch["masks"][1][b])
        outside = _check_plan(g, items, lo, hi)
        if max_outside is not None and g["R"] >= 24:
            # consistent scenes: the schedule keeps every valid tap inside the rings (nothing for the overflow list)
            assert outside <= max_outside, (outside, len(items))
        if gen == "scene" and g["R"] >= 24:
            assert len(items) <= 1.4 * g["NG"] + 12, "lock-step: about one item per row group"


def test_fan_in_count_and_limits():
    """The plan's fan-in count (sources per target pixel): 1-2 on a consistent scene, H*W when every flow points at one pixel;
    the per-pair bound of a source's contributions follows it so that a 32-bit accumulator cannot wrap."""
    from consistent_depth_amd import synthetic
    H, W = 96, 128
    g = E.geo(H, W, 2)
    b = synthetic.make_scene_batch(1, H, W, seed=4)
    f = E.fan_in(g, b["flows"][0][0], b["flows"][1][0], b["masks"][0][0], b["masks"][1][0])
    assert 1 <= f <= 12, f
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    one = np.ones((1, H, W), np.float32)
    to_point = np.stack([40.0 - xx, 30.0 - yy])          # every source samples (40, 30)
    assert E.fan_in(g, to_point, to_point, one, one) == H * W
    assert E.fan_in(g, to_point, to_point, 0 * one, 0 * one) == 0
    zoom = np.stack([-(xx - W / 2) * 0.5, -(yy - H / 2) * 0.5])   # 2x zoom-out: ~4 sources per target pixel, x 4 taps
    fz = E.fan_in(g, zoom, zoom, one, one)
    assert 8 <= fz <= 30, fz


def test_plan_degenerate_flows():
    """Empty masks, constant-target flows, pure vertical shifts beyond the ring."""
    H, W = 96, 64
    g = E.geo(H, W, 2)
    z, one = np.zeros((2, H, W), np.float32), np.ones((1, H, W), np.float32)
    yy = np.arange(H, dtype=np.float32)[:, None].repeat(W, 1)
    cases = [(z, z, 0 * one, 0 * one), (z, z, one, one)]
    up = z.copy(); up[1] = -yy                      # every source of frame 0 samples row 0 of frame 1
    down = z.copy(); down[1] = (H - 1) - yy         # ... and row H-1 the other way
    cases.append((up, down, one, one))
    shift = z.copy(); shift[1] = 60.0               # a shift larger than the ring, not mirrored by the backward flow
    cases.append((shift, shift, one, one))
    for ff, fb, mf, mb in cases:
        items, lo, hi = E.plan(g, ff, fb, mf, mb)
        _check_plan(g, items, lo, hi)


def _dist(oracle, e, ref64):
    return (abs(e["total"][0] - ref64["total"][0]) / abs(ref64["total"][0]), oracle.rel_l1(e["grad_depth"], ref64["grad_depth"]))


@pytest.mark.parametrize("name", [n for n in golden_loss_cases() if not n.startswith("nodisp")])
@pytest.mark.parametrize("cfg", [dict(pxt=2), dict(pxt=1), dict(pxt=4), dict(pxt=2, force_slow=True), dict(pxt=2, ring_rows=16),
                                 dict(pxt=2, order=1), dict(pxt=2, ring_rows=16, order=1)],
                         ids=["pxt2", "pxt1", "pxt4", "slow", "ring16", "sources_last", "ring16_sources_last"])
def test_emulated_sweep_matches_reference_goldens(oracle, name, cfg):
    batch, lr, lb, ref64, ref32 = load_loss_case(name)
    if batch["depth"].shape[-1] % cfg.get("pxt", 2):
        # a thread owns PXT ADJACENT pixels (one vector access per row): the width must be a multiple -- other widths run with
        # fewer pixels per thread here, and on the tile kernels in the product (sweep_supported)
        assert E.geo(*batch["depth"].shape[-2:], cfg["pxt"]) is None
        cfg = dict(cfg, pxt=1)
    try:
        e = E.loss(batch, lr, lb, **cfg)
    except E.FanInTooHigh:
        # the `stress` golden: flows that pile hundreds of sources onto single pixels.  Round 3's 32-bit accumulators take at most
        # 62 sources per pixel (loss_math.h): the plan kernel refuses such a pair and the product recomputes it on the exact v1
        # path (tests/test_loss_gpu.py runs this golden through the product with the sweep forced)
        g = E.geo(*batch["depth"].shape[-2:], cfg.get("pxt", 2), cfg.get("ring_rows", 0))
        assert max(E.fan_in(g, batch["flows"][0][b], batch["flows"][1][b], batch["masks"][0][b], batch["masks"][1][b])
                   for b in range(batch["depth"].shape[0])) > 62
        return
    loss_rel, grad = _dist(oracle, e, ref64)
    assert loss_rel < 1e-6
    assert grad < max(4 * oracle.rel_l1(ref32["grad_depth"], ref64["grad_depth"]), 2e-6)
    np.testing.assert_allclose(e["reprojection"], ref64["reprojection"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(e["disparity"], ref64["disparity"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("gen,kw", [("scene", {}), ("pair", {}), ("pair", {"noise_px": 40.0})], ids=["scene", "unrelated", "wild"])
@pytest.mark.parametrize("H,W", [(384, 224), (224, 384)])
def test_emulated_sweep_full_size_vs_oracle(oracle, gen, kw, H, W):
    from consistent_depth_amd import synthetic
    batch = (synthetic.make_scene_batch if gen == "scene" else synthetic.make_pair_batch)(2, H, W, seed=11, **kw)
    args = (batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], 1.0, 0.1)
    ref = oracle.consistency_loss(*args, dtype=np.float64)
    r32 = oracle.consistency_loss(*args, dtype=np.float32)
    e = E.loss(batch, 1.0, 0.1, pxt=2)
    loss_rel, grad = _dist(oracle, e, ref)
    assert loss_rel < 1e-6
    assert grad < max(4 * oracle.rel_l1(r32["grad_depth"], ref["grad_depth"]), 2e-6)
    e2 = E.loss(batch, 1.0, 0.1, pxt=2, order=1)      # rows entering / leaving an item are independent of its sources
    np.testing.assert_array_equal(e["grad_depth"], e2["grad_depth"])
    if W == 224:      # the geometry with an idle wave per frame: rows enter / leave through the service wave (round 4) -- same bits
        for order in (0, 1):
            e3 = E.loss(batch, 1.0, 0.1, pxt=2, order=order, service=True)
            np.testing.assert_array_equal(e["grad_depth"], e3["grad_depth"])
            np.testing.assert_array_equal(e["total"], e3["total"])
            assert e3["degenerate"] == e["degenerate"] and e3["overflow_entries"] == e["overflow_entries"]
    if gen == "scene" and W == 224:
        assert e["overflow_entries"] == 0 and e["slow_lanes"] < 0.01 * 2 * 2 * H * W


@pytest.mark.parametrize("mode", [1, 2])
def test_emulated_sweep_fused_heads_and_scales(oracle, mode):
    """exp / reciprocal heads, MiDaS' lambda (1e-4) and a large batch-size divisor: the accumulator counts in O(1) units
    whatever the scale of the gradient is."""
    from consistent_depth_amd import synthetic
    batch = synthetic.make_scene_batch(2, 96, 128, seed=3)
    depth = batch["depth"].astype(np.float64)
    x = np.log(depth) if mode == 1 else 1.0 / depth
    jac = depth if mode == 1 else -depth * depth
    for lr, lb in ((1.0, 0.1), (1.0, 1e-4), (0.0, 1.0)):
        ref = oracle.consistency_loss(depth, batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], lr, lb)
        r32 = oracle.consistency_loss(depth, batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], lr, lb,
                                      dtype=np.float32)
        e = E.loss(dict(batch, depth=x.astype(np.float32)), lr, lb, mode=mode, pxt=2)
        assert abs(e["total"][0] - ref["total"][0]) / abs(ref["total"][0]) < 1e-6
        assert oracle.rel_l1(e["grad_depth"], ref["grad_depth"] * jac) < max(4 * oracle.rel_l1(r32["grad_depth"], ref["grad_depth"]), 2e-6)


def test_emulated_sweep_nan_and_range_go_through_the_overflow_list(oracle):
    """Values beyond the fixed-point range and NaN are not saturated: they take the overflow list and reach the gradient."""
    from consistent_depth_amd import synthetic
    batch = synthetic.make_scene_batch(1, 64, 64, seed=2)
    batch["depth"][0, 1, 20:24, 20:24] = 1e-4      # 1/zs^2 ~ 1e8: far outside 2^17 units
    ref = oracle.consistency_loss(batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], 1.0, 0.1)
    e = E.loss(batch, 1.0, 0.1, pxt=2)
    assert e["overflow_entries"] > 0
    assert oracle.rel_l1(e["grad_depth"], ref["grad_depth"]) < 1e-5
    batch["depth"][0, 0, 5, 5] = np.nan
    e = E.loss(batch, 1.0, 0.1, pxt=2)
    assert np.isnan(e["total"][0]) and np.isnan(e["grad_depth"][0, 0, 5, 5])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_service_wave_assignment_is_bit_identical(oracle, mode):
    """Round 4: at a geometry with an idle wave per frame (W = 224, two pixels per thread) the rows entering / leaving a ring are
    handled by that wave for the whole frame (svc_load / svc_stage / svc_flush) instead of by every thread for its own columns.
    Same per-element arithmetic, so everything must come out bit for bit: all three depth heads, the slow path of wild flows, the
    overflow list of out-of-range values, the degenerate flag of a non-positive depth (which the SERVICE wave now raises: it is
    the one that sees the staged depths), NaN propagation."""
    from consistent_depth_amd import synthetic
    base = synthetic.make_scene_batch(2, 96, 224, seed=4)
    wild = synthetic.make_pair_batch(1, 64, 224, seed=5, noise_px=30.0)
    conv = {0: lambda d: d, 1: np.log, 2: lambda d: 1.0 / d}[mode]
    cases = [("scene", dict(base, depth=conv(base["depth"].astype(np.float64)).astype(np.float32))),
             ("wild", dict(wild, depth=conv(wild["depth"].astype(np.float64)).astype(np.float32)))]
    big = dict(base, depth=base["depth"].copy())
    big["depth"][0, 1, 30:34, 100:104] = 1e-4                       # contributions far outside the fixed-point range
    cases.append(("range", dict(big, depth=conv(big["depth"].astype(np.float64)).astype(np.float32))))
    bad = dict(base, depth=conv(base["depth"].astype(np.float64)).astype(np.float32))
    bad["depth"] = bad["depth"].copy()
    bad["depth"][1, 0, 50, 60] = {0: -1.0, 1: np.inf, 2: 0.0}[mode]  # a depth that is not a positive finite number -> degenerate flag
    cases.append(("degenerate", bad))
    nan = dict(base, depth=conv(base["depth"].astype(np.float64)).astype(np.float32).copy())
    nan["depth"][0, 0, 7, 9] = np.nan
    cases.append(("nan", nan))
    for name, batch in cases:
        a = E.loss(batch, 1.0, 0.1, mode=mode, pxt=2)
        b = E.loss(batch, 1.0, 0.1, mode=mode, pxt=2, service=True)
        for k in ("total", "reprojection", "disparity", "grad_depth"):
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{name} {k}")
        assert (a["degenerate"], a["overflow_entries"], a["slow_lanes"]) == (b["degenerate"], b["overflow_entries"], b["slow_lanes"]), name
        if name == "degenerate":
            assert a["degenerate"]
        if name == "range" and mode != 2:      # (the reciprocal head's units, estimated from the data, absorb this case)
            assert a["overflow_entries"] > 0


@pytest.mark.parametrize("gen,kw", [("scene", {}), ("pair", {}), ("pair", {"noise_px": 40.0})], ids=["scene", "unrelated", "wild"])
@pytest.mark.parametrize("H,W", [(384, 224), (224, 384), (96, 224)])
def test_fast_source_pass_vs_oracle(oracle, gen, kw, H, W):
    """Round 5: `process_rows_fast` (the pass the GPU runs at its compile-time geometry: premultiplied camera constants, no selects,
    one ring address per source thanks to the mirrored slot) against the fp64 oracle at the general pass's bounds -- fast path, wild
    flows (every wave votes slow: the general pass takes over inside the fast one), both execution orders, service wave on."""
    from consistent_depth_amd import synthetic
    batch = (synthetic.make_scene_batch if gen == "scene" else synthetic.make_pair_batch)(2, H, W, seed=11, **kw)
    args = (batch["depth"], batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], 1.0, 0.1)
    ref = oracle.consistency_loss(*args, dtype=np.float64)
    if ref["total"][0] == 0.0:
        pytest.skip("unrelated flows at this small size leave no pixel in bounds: empty masks, nothing to compare")
    r32 = oracle.consistency_loss(*args, dtype=np.float32)
    e = E.loss(batch, 1.0, 0.1, pxt=2, fast=True)
    loss_rel, grad = _dist(oracle, e, ref)
    assert loss_rel < 1e-6
    assert grad < max(4 * oracle.rel_l1(r32["grad_depth"], ref["grad_depth"]), 2e-6)
    np.testing.assert_allclose(e["reprojection"], ref["reprojection"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(e["disparity"], ref["disparity"], rtol=1e-6, atol=1e-7)
    e2 = E.loss(batch, 1.0, 0.1, pxt=2, order=1, fast=True)
    np.testing.assert_array_equal(e["grad_depth"], e2["grad_depth"])
    if W == 224:
        for service in (1, 2):      # 2 = the kernel's split: rows enter through the sources' own columns, leave through the service wave
            e3 = E.loss(batch, 1.0, 0.1, pxt=2, order=1, service=service, fast=True)
            np.testing.assert_array_equal(e["grad_depth"], e3["grad_depth"])
            np.testing.assert_array_equal(e["total"], e3["total"])
    g = E.loss(batch, 1.0, 0.1, pxt=2)           # the general pass on the same data: same exact-path decisions
    assert e["overflow_entries"] == g["overflow_entries"] and e["degenerate"] == g["degenerate"]
    if gen == "scene":
        assert e["slow_lanes"] == 0 and e["overflow_entries"] == 0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fast_source_pass_heads_scales_and_exact_paths(oracle, mode):
    """The fast pass with all three depth heads, MiDaS' lambda and a reprojection-free configuration; values beyond the fixed-point range,
    a degenerate depth and NaN take the same exact paths as in the general pass."""
    from consistent_depth_amd import synthetic
    base = synthetic.make_scene_batch(2, 96, 224, seed=4)
    depth = base["depth"].astype(np.float64)
    conv = {0: lambda d: d, 1: np.log, 2: lambda d: 1.0 / d}[mode]
    jac = {0: 1.0, 1: depth, 2: -depth * depth}[mode]
    for lr, lb in ((1.0, 0.1), (1.0, 1e-4), (0.0, 1.0)):
        ref = oracle.consistency_loss(depth, base["flows"], base["masks"], base["intrinsics"], base["extrinsics"], lr, lb)
        r32 = oracle.consistency_loss(depth, base["flows"], base["masks"], base["intrinsics"], base["extrinsics"], lr, lb, dtype=np.float32)
        e = E.loss(dict(base, depth=conv(depth).astype(np.float32)), lr, lb, mode=mode, pxt=2, fast=True, service=True)
        assert abs(e["total"][0] - ref["total"][0]) / abs(ref["total"][0]) < 1e-6
        assert oracle.rel_l1(e["grad_depth"], ref["grad_depth"] * jac) < max(4 * oracle.rel_l1(r32["grad_depth"], ref["grad_depth"]), 2e-6)
        assert e["slow_lanes"] == 0
    big = dict(base, depth=base["depth"].copy())
    big["depth"][0, 1, 30:34, 100:104] = 1e-4                       # contributions far outside the fixed-point range
    big = dict(big, depth=conv(big["depth"].astype(np.float64)).astype(np.float32))
    a, b = E.loss(big, 1.0, 0.1, mode=mode, pxt=2), E.loss(big, 1.0, 0.1, mode=mode, pxt=2, fast=True)
    assert a["overflow_entries"] == b["overflow_entries"] and (mode == 2 or b["overflow_entries"] > 0)
    assert oracle.rel_l1(b["grad_depth"], a["grad_depth"]) < 2e-6
    bad = dict(base, depth=conv(depth).astype(np.float32).copy())
    bad["depth"][1, 0, 50, 60] = {0: -1.0, 1: np.inf, 2: 0.0}[mode]
    assert E.loss(bad, 1.0, 0.1, mode=mode, pxt=2, fast=True)["degenerate"]
    nan = dict(base, depth=conv(depth).astype(np.float32).copy())
    nan["depth"][0, 0, 7, 9] = np.nan
    e = E.loss(nan, 1.0, 0.1, mode=mode, pxt=2, fast=True)
    assert np.isnan(e["total"][0]) and np.isnan(e["grad_depth"][0, 0, 7, 9])


def test_mirrored_slot_survives_forced_general_pass():
    """Slot R of a ring mirrors slot 0 (ring_rows): rows staged / flushed by either assignment keep both copies in step, and a run that
    mixes fast and general passes (force_slow: every wave votes slow inside the fast pass) drains every accumulator word."""
    from consistent_depth_amd import synthetic
    batch = synthetic.make_scene_batch(1, 96, 224, seed=9)
    a = E.loss(batch, 1.0, 0.1, pxt=2)
    b = E.loss(batch, 1.0, 0.1, pxt=2, fast=True, force_slow=True, service=2)         # (the emulation fails on a non-drained accumulator)
    np.testing.assert_array_equal(a["grad_depth"], b["grad_depth"])
    np.testing.assert_array_equal(a["total"], b["total"])


@pytest.mark.parametrize("gen,kw", [("scene", {}), ("pair", {"noise_px": 40.0})], ids=["scene", "wild"])
def test_plan_flag_all_taps_inside_changes_nothing(gen, kw):
    """Rec::inw (end of round 5): the planner marks the items whose valid sources all sample usable ring rows; the fast pass then skips
    the window clamp and the slow-path vote.  With the flag withheld the clamp is the identity for those lanes and mask-0 lanes contribute
    exact zeros from whatever resident row they read: gradients and losses must be BIT-identical with and without it.  Scene flows claim
    the flag on (almost) every item, wild flows on few -- and the items that do not claim it still take the clamped / voted route."""
    from consistent_depth_amd import synthetic
    H, W = 384, 224
    batch = (synthetic.make_scene_batch if gen == "scene" else synthetic.make_pair_batch)(2, H, W, seed=21, **kw)
    on = E.loss(batch, 1.0, 0.1, pxt=2, fast=True, service=2)
    off = E.loss(batch, 1.0, 0.1, pxt=2, fast=True, service=2, inw=False)
    for k in ("grad_depth", "total", "reprojection", "disparity"):
        np.testing.assert_array_equal(on[k], off[k])
    assert on["overflow_entries"] == off["overflow_entries"] and on["slow_lanes"] == off["slow_lanes"]
    g = E.geo(H, W, 2)
    flags = E.inw_flags(g, batch["flows"][0][0], batch["flows"][1][0], batch["masks"][0][0], batch["masks"][1][0])
    items, lo, hi = E.plan(g, batch["flows"][0][0], batch["flows"][1][0], batch["masks"][0][0], batch["masks"][1][0])
    with_group = items[:, :2] >= 0
    assert not flags[~with_group].any()                    # only items with a group can claim it
    share = flags[with_group].mean()
    if gen == "scene":
        assert share > 0.9, share
    else:
        assert share < 0.9, share
