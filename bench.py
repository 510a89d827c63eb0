#!/usr/bin/env python3
"""Benchmark of the fine-tuning hot path on MI355X (contract: task prompt section 4).

A "step" = one optimisation step of depth_fine_tuning.py's loop body on one batch of BS=4
frame pairs per GPU of a synthetic 384x224 clip: batch gather from the HBM-resident pair store (one HIP launch),
hourglass forward (train-mode BN, 8 images), fused HIP geometric-consistency loss (+ analytic backward), CNN backward,
[RCCL all-reduce], HIP Adam.  The clip (BASELINE configs[2]: 244 frames, hierarchical pair sampling -> 715 pairs) is
uploaded once before the timed region; every step takes the next batch of the epoch's shared-seed permutation, sharded
over the ranks exactly like the training driver does (parallel.shard_indices).  value = frame pairs / second over all ranks.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...            (re-launches itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W

Extra objects in the JSON line:
  roofline          fused loss kernel at an HBM-saturating launch (--loss-batch pairs, working set
                    well beyond the 256 MB Infinity Cache), HIP events on the launch stream
  roofline_in_step  the same kernel as launched inside the timed steps (B = 4: 13.8 MB, cache
                    resident and latency bound -- reported for honesty, not an HBM measurement)
  roofline_conv     the convolutions of the step against the matrix-core roof: algorithmic conv flops of the step (forward +
                    input gradient + weight gradient = 3 x forward) / the measured step time
  cpu_baseline      the reference step restated on the host CPU (oracle/cpu_step.py), rank 0, N=1 (mc only)

    python bench.py --model midas2 --height 384 --width 384 --batch-size 8      BASELINE configs[4] shape on one GPU
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
LOSS_BYTES_PER_PAIR_PX = 10 * 4  # read depth x2, flow x4, mask x2, write grad x2 (fp32), SURVEY.md 8d
# `roofline.traffic` is MEASURED IN THIS RUN (rank 0, N = 1): two bounded `rocprofv3 --pmc` subprocess passes (FETCH_SIZE, then
# WRITE_SIZE -- they do not fit one pass; no trace option next to them) over tools/loss_bench.py at the same launch size, read back
# from the rocpd database, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB and,
# on gfx950, FETCH_SIZE reports half the bytes of a wide coalesced streaming read (x 2).  null when rocprofv3 is unavailable or a
# pass fails -- never a number copied from an older build (round 3 printed a constant from profiles/).
# Matrix-core roofs (TFLOP/s, dense; /opt/skills/guides/MI355X_MICROARCH.md).  The split-operand convolutions compute one fp32
# result from SIX bf16 products, so the roof they run against, in fp32-equivalent flops, is the dense BF16 peak / 6.
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP32_PEAK_TFLOPS = 157.3


def mc_conv_macs(H, W):
    """Forward multiply-accumulates of all convolutions of the mc hourglass for ONE HxW image (SURVEY.md section 8d:
    52.83 GMAC at 384x224), walked over the architecture tables of consistent_depth_amd/monodepth/hourglass.py."""
    from consistent_depth_amd.monodepth import hourglass as HG
    total = H * W * 3 * 49 * 128 + H * W * 64 * 9          # stem 7x7 3->128, head 3x3 64->1 (the unused confidence head is not run)

    def inception(kind, h, w):
        cin, cfg = HG.INCEPTION[kind]
        n = h * w * cin * cfg[0][0]
        for k, mid, out in cfg[1:]:
            n += h * w * (cin * mid + mid * k * k * out)
        return n

    def channels(level, h, w):
        n = 0
        for side in HG.CHANNELS[level]:
            hh, ww = h, w
            for it in side:
                if it == "pool":
                    hh, ww = hh // 2, ww // 2
                elif it == "up":
                    hh, ww = hh * 2, ww * 2
                elif isinstance(it, tuple):
                    n += channels(it[1], hh, ww)
                else:
                    n += inception(it, hh, ww)
        return n
    return total + channels(4, H, W)


def module_conv_macs(net, x):
    """Forward MACs of every nn.Conv2d of `net` for the input `x` (hooks; used for the MiDaS-shaped backbone)."""
    macs, hooks = [0], []

    def hook(m, inp, out):
        macs[0] += out.numel() * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        net(x)
    for h in hooks:
        h.remove()
    return macs[0]


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line is the only thing printed on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="mc", choices=["mc", "midas2"],
                    help="depth model plugin: mc = Mannequin-Challenge hourglass (BASELINE configs[2]/[3], the headline); midas2 = the "
                         "MiDaS-v2-shaped ResNeXt-101 backbone (configs[4]: use --height 384 --width 384 --batch-size 8)")
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--backend", default=os.environ.get("CD_AMD_MC_BACKEND", "hip"), choices=["torch", "hip"],
                    help="convolutions: hip = hand-written gfx950 engine (BASELINE configs[2]); "
                         "torch = PyTorch-ROCm/MIOpen (configs[1], ~2 min of MIOpen start-up)")
    ap.add_argument("--frames", type=int, default=0,
                    help="frames of the synthetic clip, resident on every GPU.  0 = by --gpus: 244 (BASELINE configs[2]: 715 pairs) on one GPU, "
                         "1000 (BASELINE configs[3]: 2979 pairs, 7.2 GB per GPU) with --gpus > 1")
    ap.add_argument("--generator", choices=["auto", "host", "device"], default="auto",
                    help="synthetic clip: numpy on the host (the clip every single-GPU line of rounds 1-4 ran on) or torch on the device "
                         "(same recipe, same cameras and surface, its own noise / mask / colour stream; seconds instead of minutes for 1000 "
                         "frames).  auto = host for <= 244 frames, device beyond")
    ap.add_argument("--max-pairs", type=int, default=0, help="keep only the first N pairs of the clip (quick runs)")
    ap.add_argument("--loss-batch", type=int, default=256,
                    help="pairs per launch of the roofline micro-benchmark (256 = one pair per CU, 0.88 GB: the launch every round has "
                         "quoted; a second, 4x larger launch is reported next to it as roofline.sustained)")
    ap.add_argument("--loss-iters", type=int, default=40)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("CD_AMD_STEP_GRAPH", "1")),
                    help="1: replay the step from a HIP graph after the eager warm-up steps (GraphedFineTuneStep); 0: eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the short BASELINE configs[4] side measurement (midas2 backbone, 384x384, BS8) that the default N=1 run appends")
    ap.add_argument("--config5-timeout", type=int, default=150)
    ap.add_argument("--no-loss-microbench", action="store_true")
    ap.add_argument("--no-loss-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--traffic-timeout", type=int, default=90, help="seconds each PMC pass may take")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed full steps of the CPU baseline (median reported; + 10 loss-only evaluations)")
    ap.add_argument("--cpu-timeout", type=int, default=240, help="seconds the CPU-baseline subprocess may take")
    ap.add_argument("--miopen-find", action="store_true",
                    help="torch backend only: let MIOpen benchmark-search every conv (minutes of start-up)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run on this node."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks on port {port}")
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


class EpochPlans:
    """The training driver's batch schedule (depth_fine_tuning.py): per epoch a shared-seed permutation of the pair list cut
    into global batches, this rank's slice of each, uploaded once per epoch."""

    def __init__(self, n_pairs, rank, world, batch_size, device, seed=0):
        self.n, self.rank, self.world, self.bs, self.device, self.seed = n_pairs, rank, world, batch_size, device, seed
        self.epoch, self.pos, self.plan = -1, 0, []

    def next(self):
        from consistent_depth_amd import parallel
        while self.pos >= len(self.plan):
            self.epoch += 1
            plan = parallel.shard_indices(self.n, self.epoch, self.seed, self.rank, self.world, self.bs)
            plan = [ids for ids in plan if len(ids) == self.bs]      # full batches only: one graph signature
            self.plan, self.pos = parallel.plan_to_device(plan, self.device), 0
        ids = self.plan[self.pos]
        self.pos += 1
        return ids


def profile_collect(lib, cap):
    ms = (ctypes.c_float * cap)()
    bs = (ctypes.c_int * cap)()
    n = ctypes.c_int(0)
    rc = lib.cd_profile_end(ms, bs, cap, ctypes.byref(n))
    assert rc == 0
    return np.array(ms[:n.value]), np.array(bs[:n.value])


LOSS_WARM_CALLS = 100


def midas_path_note():
    """What the configs[4] line ran on."""
    return ("every convolution on hand-written HIP kernels, forward / input gradient / weight gradient: k >= 3 incl. the grouped 32x8d 3x3 (1/3 of "
            "the multiply-adds) and the dense 1x1 (2/3 of the multiply-adds; >= 512 channels: the chunked kernel conv1x1_split_kc_kernel and "
            "wgrad1x1_split, 96x96 planes: the LDS-resident 1x1 kernel) in split-bf16 arithmetic, the RGB stem and the 1-channel head on the fp32 "
            "matrix instruction; no library GEMM; BatchNorm (+ identity) (+ ReLU), ReLU, adds, max-pool: hand-written blocks (csrc/bn_block.hip); "
            "bilinear x2: hand-written kernels; stride 2 (k >= 3) = stride 1 + sub-sampling; loss + Adam: hand-written HIP)")


def loss_microbench(lib, B, H, W, iters, device, warm=LOSS_WARM_CALLS):
    """Fused loss kernel at an HBM-saturating batch: per-launch ms from HIP events on the stream."""
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.loss import consistency_loss as CL
    base = synthetic.make_scene_batch(8, H, W, seed=99)
    rep = (B + 7) // 8
    t = lambda a: torch.tensor(a, device=device).repeat((rep,) + (1,) * (a.ndim - 1))[:B].contiguous()  # noqa: E731
    depth = torch.log(t(base["depth"]))
    # de-duplicate content a little so repeated pairs are not bit-identical
    depth += 0.01 * torch.randn_like(depth)
    flows, masks = [t(f) for f in base["flows"]], [t(m) for m in base["masks"]]
    intr, extr = t(base["intrinsics"]), t(base["extrinsics"])
    msum, twin = CL.mask_sums(masks[0], masks[1]), CL.tile_windows(flows, masks)
    depth.requires_grad_(True)
    call = lambda: CL.consistency_loss(depth, flows, masks, intr, extr, 1.0, 0.1, mask_sums=msum, depth_mode=CL.DEPTH_EXP,  # noqa: E731
                                       tile_windows=twin)
    # untimed warm-up calls: plans, workspace -- and the clocks.  A cold series of 0.2 ms calls starts fast, sags for ~20 calls and settles
    # (profiles/loss_sweep_variants_r05.txt: 3 warm-up calls 0.2347 ms, 100 calls 0.2082 ms on the same box); ~20 ms of calls reach the
    # settled state the timed calls then stay in.
    for _ in range(warm):
        call()
    torch.cuda.synchronize()
    assert lib.cd_profile_begin(iters) == 0
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    ms, _ = profile_collect(lib, iters)
    return ms


def loss_traffic_pmc(B, H, W, timeout, kernel_substr="loss_sweep_kernel"):
    """HBM bytes per launch of the loss gradient kernel from rocprofv3 PMC counters, measured now on this GPU: one subprocess pass
    per counter (`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, nothing else on the command line) around tools/loss_bench.py at
    B pairs.  Returns (bytes or None, note)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [rp, "--pmc", counter, "-d", td, "-o", "pmc", "--", sys.executable, os.path.join(REPO, "tools", "loss_bench.py"),
                   "--batches", str(B), "--iters", "3", "--height", str(H), "--width", str(W)]
            try:
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, f"{counter} pass exceeded {timeout}s"
            per_dispatch = {}
            for db in glob.glob(os.path.join(td, "**", "*.db"), recursive=True):
                try:
                    rows = sqlite3.connect(db).execute("select name, counter_name, counter_value, dispatch_id from pmc_events").fetchall()
                except sqlite3.Error:
                    continue
                for name, cn, cv, did in rows:       # counters come per instance (XCD): sum per dispatch
                    if cn == counter and kernel_substr in name:
                        per_dispatch[did] = per_dispatch.get(did, 0.0) + cv
            if not per_dispatch:
                return None, f"no {kernel_substr} dispatch in the {counter} pass"
            got[counter] = 1024.0 * sum(per_dispatch.values()) / len(per_dispatch)
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], "rocprofv3 --pmc, separate passes, this run: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes"


def main():
    args = parse()
    from consistent_depth_amd import _native, parallel
    from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
    from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank, local_rank, world = parallel.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    device = parallel.local_device(local_rank)
    torch.cuda.set_device(device)
    lib = _native.lib()
    B, H, W = args.batch_size, args.height, args.width

    model_cls = get_depth_model(args.model)
    # the reference takes learning rate and lambda_view_baseline from the model adapter (params.py:110-119)
    params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=model_cls.lambda_view_baseline, lambda_parameter=0,
                                learning_rate=model_cls.learning_rate, optimizer="Adam")
    # the reference sets cudnn.benchmark=True (depth_fine_tuning.py:220-221); MIOpen's exhaustive
    # find over the 157 conv shapes x {fwd, dgrad, wgrad} takes many minutes, so the default here
    # is MIOpen's immediate (heuristic) mode
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    log(f"rank {rank}/{world} building {args.model} model, conv backend {args.backend}")
    model = model_cls(backend=args.backend, seed=0)
    model.train()
    if args.model == "midas2" and args.graph and os.environ.get("CD_AMD_MIDAS_GRAPH", "1") == "0":
        log("midas2: CD_AMD_MIDAS_GRAPH=0 -> eager steps")
        args.graph = 0
    eager_step = FineTuneStep(model, params, world=world)
    step = GraphedFineTuneStep(eager_step, eager_steps=max(1, min(2, args.warmup - 1))) if args.graph else eager_step
    # the clip, resident in HBM on every rank (replicated like the reference's dataset on every DataLoader worker)
    from consistent_depth_amd.loaders.pair_store import PairStore
    if not args.frames:
        args.frames = 244 if world == 1 else 1000
    gen_dev = args.generator == "device" or (args.generator == "auto" and args.frames > 244)
    t_gen = time.perf_counter()
    store = (PairStore.synthetic_device if gen_dev else PairStore.synthetic)(args.frames, H, W, seed=0, device=device, max_pairs=args.max_pairs or None)
    torch.cuda.synchronize()
    plans = EpochPlans(len(store), rank, world, B, device, seed=0)
    log(f"pair store: {args.frames} frames, {len(store)} pairs, {store.nbytes / 1e9:.2f} GB resident, generated on the "
        f"{'device' if gen_dev else 'host'} in {time.perf_counter() - t_gen:.1f}s")
    scene_scale = 1.0
    if args.model == "midas2":
        # The reference's scale-calibration stage (scale_calibration.py:305-313) rescales the camera translations so that the
        # geometry agrees with the network's INITIAL depth; the synthetic scene is in [0.5, 4] while a MiDaS-style network
        # predicts inverse depth of order 1e2..1e3.  Same here: median initial depth / median scene depth, decided by rank 0.
        with torch.no_grad():
            sel = torch.arange(0, store.color.shape[0], max(1, store.color.shape[0] // 8), device=device)[:8]
            d0 = model.estimate_depth(store.color[sel])
            gt = torch.as_tensor(store.gt_depth[sel.cpu().numpy()], dtype=torch.float32, device=device)
            factor = (d0.median() / gt.median()).reshape(1)
        if world > 1:
            dist.broadcast(factor, 0)
        scene_scale = float(factor.item())
        store.scale_scene_(scene_scale)
        log(f"scale calibration: scene scaled by {scene_scale:.5g} to the network's initial depth")

    step_losses = []

    def run(n):
        last = None
        for _ in range(n):
            last, _, _ = step.step_from_store(store, plans.next())
            step_losses.append(last)
        return last

    for i in range(args.warmup):
        run(1)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if args.graph and getattr(step, "graphed", None) is None:
        # fewer warm-up steps than the capture needs (eager steps + 1): finish the one-time capture outside the timed region
        for i in range(4):
            run(1)
            torch.cuda.synchronize()
            if step.graphed is not None:
                break
        log(f"graph capture finished in {i + 1} extra untimed step(s)")
    graphed = bool(args.graph) and getattr(step, "graphed", None) is True
    if args.graph and not graphed:
        log(f"HIP graph capture not active ({getattr(step, 'capture_error', None) or 'needs >= 3 warm-up steps'}); eager steps")
    if not graphed:   # event records of the in-step loss profiler are not captured into graphs: eager only
        assert lib.cd_profile_begin(args.steps + 8) == 0
    from consistent_depth_amd import parallel as _par
    if world > 1:     # HIP events around every step's gradient all-reduce: the first real multi-GPU run diagnoses itself
        _par.exchange_timer.start(args.steps)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_losses.clear()
    last_loss = run(args.steps)
    t_enqueue = time.perf_counter() - t0   # host time to enqueue all steps (no sync inside the loop)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    exchange_ms = _par.exchange_timer.stop() if world > 1 else []
    # A non-finite loss makes the device-side guard skip the Adam update of that step (the reference skips NaN steps too,
    # depth_fine_tuning.py:372-376): such a step is not a full step, and a timed region containing one is not a measurement.
    # (graph replays return a clone of the static loss buffer: one tensor per step here too)
    finite = torch.isfinite(torch.stack([l.reshape(()) for l in step_losses])).sum().float().reshape(1)
    if world > 1:     # every rank takes the same decision (a rank leaving alone would hang the others in the next collective)
        dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    finite_steps = int(finite.item())
    if finite_steps != args.steps:
        sys.exit(f"bench.py: {args.steps - finite_steps} of {args.steps} timed steps had a non-finite loss (Adam update skipped): "
                 f"not a valid measurement")
    # What the HOST does per step, measured with the GPU idle at the start of each step (5 extra steps, every rank): t_enqueue above is the
    # wall time of the free-running loop, in which hipGraphLaunch blocks while the previous replay of the same graph is still executing
    # (round 6, tools/host_cost.py: 11.3 ms/step free-running vs 1.3-1.4 ms/step of actual host work on both clips).
    host_idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step.step_from_store(store, plans.next())
        host_idle.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    # With graph replay the in-step loss kernel is timed in 3 eager steps after the timed region.  Those steps contain the
    # gradient all-reduce, so the decision to run them must be the same on every rank (capture could fail on a single one).
    extra = graphed
    if world > 1:
        flag = torch.tensor([1.0 if graphed else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        extra = flag.item() > 0
    if extra:
        if graphed:       # (an eager rank has had the profiler on since before the timed region)
            assert lib.cd_profile_begin(16) == 0
        for i in range(3):
            eager_step.step_from_store(store, plans.next())
        torch.cuda.synchronize()
    ms_step, _ = profile_collect(lib, 16 if graphed else args.steps + 8)
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    pairs_per_s = B * world * args.steps / elapsed
    if args.model == "mc":
        macs_per_image = mc_conv_macs(H, W)
    else:
        macs_per_image = module_conv_macs(model.model, torch.zeros(1, 3, H, W, device=device))
    log(f"timed region: {args.steps} steps in {elapsed:.3f}s -> {pairs_per_s:.2f} pairs/s "
        f"(free-running host loop {1e3 * t_enqueue / args.steps:.1f} ms/step)")

    out = {
        "metric": "frame-pairs/sec fine-tuning @384x224 BS4; warp+loss HBM GB/s vs peak",
        "value": round(pairs_per_s, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"mc hourglass (random init, seed 0) test-time fine-tuning steps over a synthetic {args.frames}-frame "
                                f"{H}x{W} clip ({len(store)} pairs, hierarchical sampling, HBM-resident pair store, shared-seed shards; full "
                                f"batches only -- the epoch's short last batch is left out so that ONE graph signature is timed), "
                                f"BS{B} pairs/GPU, lambda_r 1.0 lambda_b 0.1, Adam lr 4e-4 ("
                                + (("BASELINE configs[3]: " if args.frames == 1000 else "BASELINE configs[2]'s clip resident on every GPU, "
                                    "configs[3]'s data parallelism -- the literal configs[3] clip is --frames 1000: ")
                                   + f"DP over {world} GPUs, one flat RCCL all-reduce of the gradients per step, " if world > 1 else
                                   ("BASELINE configs[2]: " if args.frames == 244 and not args.max_pairs else "")
                                   if args.backend == "hip" else "BASELINE configs[1]: ")
                                + ("full HIP conv+loss path)" if args.backend == "hip" else "HIP loss+Adam, convs on PyTorch-ROCm/MIOpen)"))
                               if args.model == "mc" else
                               (f"midas2 plugin: MiDaS-v2-shaped backbone (ResNeXt-101 32x8d + feature-fusion decoder restated, random init, "
                                f"seed 0), fine-tuning steps over a synthetic {args.frames}-frame {H}x{W} clip ({len(store)} pairs), BS{B} pairs/GPU, "
                                f"lambda_r 1.0 lambda_b 1e-4, Adam lr 1e-4 (BASELINE configs[4] shape on {world} GPU(s); "
                                + (midas_path_note() if args.backend == "hip" else "convolutions: PyTorch-ROCm/MIOpen)")),
                   "model": args.model,
                   "conv_backend": args.backend,
                   "conv_arith": ("fp32 results from split operands: every fp32 input = 3 exact bf16 terms, 6 cross products on the BF16 matrix "
                                  "cores, fp32 accumulate (all convolutions but the RGB stem, fwd/dgrad/wgrad; as close to fp64 as the fp32 MFMA: profiles/mfma_split_exp_r02.txt, "
                                  "tests/test_conv_gpu.py); RGB stem and small-image 1x1 on the fp32 MFMA" if lib.cd_get_conv_arith() >= 1 else "fp32 MFMA (CD_AMD_CONV_ARITH=fp32)")
                   if args.backend == "hip" else "MIOpen fp32",
                   "clip_generator": "device (torch, PairStore.synthetic_device)" if gen_dev else "host (numpy, PairStore.synthetic)",
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "hip_graph": graphed,
                   **({"dp_exchange": ("all-reduce + Adam inside the step graph (CD_AMD_DP_GRAPH_COLLECTIVE=1)" if getattr(step, "graph_collective", False)
                                       else "one flat all-reduce of [gradients | loss] + one Adam launch per step, eager, after the graph replay"),
                       "dp_backend": dist.get_backend(), "dp_ranks": dist.get_world_size(),
                       "dp_payload_mb": round(eager_step.opt.reduce_buffer.numel() * 4 / 1e6, 2),
                       # rank 0's HIP events around the collective of every timed step (enqueue -> completion as the step sees it: it
                       # includes waiting for the slowest rank to arrive); empty when the collective sits inside the step graph
                       "dp_exchange_ms": ({"mean": round(float(np.mean(exchange_ms)), 4), "min": round(float(np.min(exchange_ms)), 4),
                                           "max": round(float(np.max(exchange_ms)), 4), "steps": len(exchange_ms)} if exchange_ms else None)}
                      if world > 1 else {}),
                   "host_ms_per_step": round(1e3 * float(np.median(host_idle)), 3),
                   "host_ms_per_step_note": "host time of one step (gather launch + graph replay + bookkeeping) with the GPU idle at its start, median of 5 extra steps",
                   "host_loop_ms_per_step": round(1e3 * t_enqueue / args.steps, 2),
                   "host_loop_note": "wall time of the free-running timed loop per step: includes hipGraphLaunch blocking while the previous replay runs (not host work)",
                   "last_loss": float(last_loss.item()), "finite_loss_steps": finite_steps,
                   **({"scene_scale": round(scene_scale, 6)} if scene_scale != 1.0 else {})},
    }

    if rank == 0:
        px = H * W
        flops_per_pair = 2 * 3 * 2 * macs_per_image            # 2 images x (forward + input gradient + weight gradient) x 2 flop/MAC
        ach_tf = flops_per_pair * pairs_per_s / world / 1e12  # per GPU
        # (midas2 since round 6: its dense 1x1 convolutions run on the split-operand kernels too -- every convolution but the RGB stem and the
        # 1-channel head is priced against the split roof)
        split = args.backend == "hip" and args.model in ("mc", "midas2") and lib.cd_get_conv_arith() >= 2
        peak_tf = MFMA_BF16_PEAK_TFLOPS / 6 if split else MFMA_FP32_PEAK_TFLOPS
        out["roofline_conv"] = {
            "bound": "mfma", "achieved": round(ach_tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": round(ach_tf / peak_tf, 4),
            "flops_per_pair": flops_per_pair, "per": "GPU, whole step time (BatchNorm, loss, Adam and launch gaps included: a lower bound "
                                                     "of what the convolution kernels reach while they run)",
            "peak_note": ("fp32-equivalent roof of the split-operand kernels = dense BF16 peak 2500 / 6 products" if split else
                          "fp32 matrix instruction (v_mfma_f32_16x16x4_f32)"),
            "frac_of_fp32_mfma_peak": round(ach_tf / MFMA_FP32_PEAK_TFLOPS, 4),
            "mfma_busy_source": "profiles/rocprofv3_bench_pmc_r06.txt (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES per kernel family), "
                                "profiles/conv_roofline_r06.txt (per launch)"}
        in_step_ms = float(np.mean(ms_step)) if len(ms_step) else None
        if in_step_ms:
            ach = LOSS_BYTES_PER_PAIR_PX * px * B / (in_step_ms * 1e-3) / 1e9
            out["roofline_in_step"] = {"kernel": "loss_source_kernel + loss_gather4_kernel (one gradient launch)", "bound": "hbm", "achieved": round(ach, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                       "traffic": None, "launch_pairs": B, "avg_ms": round(in_step_ms, 5),
                                       "note": "13.8 MB per launch: cache resident, launch-latency bound"}
        if not args.no_loss_microbench:
            log("loss kernel micro-benchmark")
            try:
                ms = loss_microbench(lib, args.loss_batch, H, W, args.loss_iters, device)
            except (RuntimeError, AssertionError) as e:    # never lose the headline line to the side measurement
                log(f"loss micro-benchmark failed: {type(e).__name__}: {e}")
                ms = []
        if not args.no_loss_microbench and len(ms):
            avg = float(np.mean(ms))
            ach = LOSS_BYTES_PER_PAIR_PX * px * args.loss_batch / (avg * 1e-3) / 1e9
            sweep = (H, W) == (384, 224) and args.loss_batch >= 96     # the default dispatch of cd_consistency_loss_fwd_bwd (loss_sweep.hip::sweep_preferred)
            traffic, traffic_note = (None, "not measured (--no-loss-traffic)")
            if sweep and world == 1 and not args.no_loss_traffic:
                log("loss kernel HBM traffic (2 rocprofv3 --pmc passes)")
                traffic, traffic_note = loss_traffic_pmc(args.loss_batch, H, W, args.traffic_timeout)
            alg = LOSS_BYTES_PER_PAIR_PX * px * args.loss_batch
            out["roofline"] = {"kernel": "loss_sweep_kernel = the WHOLE gradient call (row sweep, one workgroup per pair: per-pair constants, sweep, overflow "
                                         "entries, per-pair losses and their mean in ONE kernel + the 4-byte counter reset in front; the HIP events bracket "
                                         "everything cd_consistency_loss_fwd_bwd enqueues)" if sweep else
                                         "loss_source_kernel + loss_gather4_kernel (one gradient launch)",
                               "bound": "hbm", "achieved": round(ach, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": traffic_note,
                               "traffic_over_algorithmic": round(traffic / alg, 4) if traffic else None,
                               "launch_pairs": args.loss_batch, "avg_ms": round(avg, 5), "median_ms": round(float(np.median(ms)), 5),
                               "min_ms": round(float(np.min(ms)), 5), "max_ms": round(float(np.max(ms)), 5), "calls": int(len(ms)),
                               "warmup_calls": LOSS_WARM_CALLS,
                               "algorithmic_bytes_per_launch": alg,
                               "lib": lib.cd_build_info().decode()}
            if sweep:
                # The row sweep gives every CU ONE pair at a time; at 256 pairs = 256 CUs the launch lasts as long as its slowest pair
                # (plans differ in length) -- with several pairs per CU the workgroups balance.  The same kernel at 4x the pairs:
                try:
                    ms4 = loss_microbench(lib, 4 * args.loss_batch, H, W, max(4, args.loss_iters // 2), device, warm=LOSS_WARM_CALLS // 4)
                    avg4 = float(np.mean(ms4))
                    ach4 = LOSS_BYTES_PER_PAIR_PX * px * 4 * args.loss_batch / (avg4 * 1e-3) / 1e9
                    out["roofline"]["sustained"] = {"launch_pairs": 4 * args.loss_batch, "avg_ms": round(avg4, 5), "achieved": round(ach4, 1),
                                                    "frac": round(ach4 / HBM_PEAK_GBS, 4)}
                except (RuntimeError, AssertionError) as e:
                    log(f"4x loss micro-benchmark failed: {type(e).__name__}: {e}")
        if world == 1 and not args.no_cpu_baseline and args.model == "mc":
            # the reference step restated on the host (oracle/cpu_step.py), in a bounded subprocess so a slow
            # host can never stall the benchmark: 1 warm-up + --cpu-steps timed steps of the same BS4 workload
            log("cpu baseline")
            import pickle
            import subprocess
            import tempfile
            cores = usable_cores()
            ids0 = EpochPlans(len(store), 0, 1, B, device, seed=0).next()
            im0, meta0 = store.batch(ids0)
            g0 = meta0["geometry_consistency"]
            images_np = im0.cpu().numpy()
            b_np = {"flows": [f.cpu().numpy() for f in g0["flows"]], "masks": [m.cpu().numpy() for m in g0["masks"]],
                    "intrinsics": meta0["intrinsics"].cpu().numpy(), "extrinsics": meta0["extrinsics"].cpu().numpy()}
            with tempfile.TemporaryDirectory() as td:
                blob = os.path.join(td, "in.pkl")
                with open(blob, "wb") as f:
                    pickle.dump({"state": {k: v.detach().cpu() for k, v in model.netG.state_dict().items()},
                                 "images": images_np, "batch": b_np, "steps": args.cpu_steps, "threads": cores}, f)
                code = ("import pickle,sys,torch;sys.path.insert(0,%r);from oracle import cpu_step;"
                        "d=pickle.load(open(%r,'rb'));"
                        "f,l,a=cpu_step.time_steps(d['state'],d['images'],d['batch'],n_steps=d['steps'],warmup=1,threads=d['threads'],loss_steps=10);"
                        "print('SEC',f,l,min(a),max(a))"
                        % (REPO, blob))
                env = dict(os.environ, OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
                try:
                    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
                    sec, loss_sec, smin, smax = [float(x) for x in [ln for ln in r.stdout.splitlines() if ln.startswith("SEC")][-1].split()[1:5]]
                    out["cpu_baseline"] = {"value": round(B / sec, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                           "loss_only": {"value": round(B / loss_sec, 3), "unit": "frame-pairs/s", "ms_per_call": round(loss_sec * 1e3, 2),
                                                         "sample": f"10 evaluations (+1 warm-up) of loss + d loss / d depth on the same BS{B} batch, C oracle "
                                                                   "(oracle/cd_oracle.c, fp32, single-threaded), median"},
                                           "sample": f"{args.cpu_steps} full steps (+1 warm-up), MEDIAN per-step time extrapolated to pairs/s, on the first "
                                                     f"BS{B} batch of the same clip: torch CPU fp32 hourglass fwd+bwd (train-mode BN) + "
                                                     f"C-oracle loss + torch Adam, {sec:.2f} s/step (min {smin:.2f}, max {smax:.2f}), {cores} threads = the "
                                                     f"cgroup's CPU quota"}
                except (subprocess.TimeoutExpired, IndexError, ValueError) as e:
                    out["cpu_baseline"] = {"value": None, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                                           "sample": f"not completed within {args.cpu_timeout}s ({type(e).__name__})"}
        if world == 1 and args.model == "mc" and not args.no_config5:
            # BASELINE configs[4] on this GPU, as a SIDE measurement in its own process (its own line is `python bench.py --model
            # midas2 --height 384 --width 384 --batch-size 8`): 5 timed steps of the midas2 plugin, bounded by a timeout
            log("config 5 side measurement (midas2, 384x384, BS8)")
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--model", "midas2", "--height", "384", "--width", "384", "--batch-size", "8",
                   "--steps", "5", "--warmup", "2", "--frames", "20", "--no-loss-microbench", "--no-cpu-baseline", "--backend", args.backend]
            stderr_tail = ""
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.config5_timeout)
                stderr_tail = (r.stderr or "")[-300:]      # (a run that refuses its own timed region -- non-finite losses -- says so here)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                c5 = json.loads(line)
                out["config5"] = {"value": c5["value"], "unit": c5["unit"], "ms_per_step": c5["ms_per_step"], "steps": c5["steps"],
                                  "n_gpus": 1, "finite_loss_steps": c5["config"].get("finite_loss_steps"),
                                  "last_loss": c5["config"].get("last_loss"), "hip_graph": c5["config"].get("hip_graph"),
                                  "workload": c5["config"]["workload"], "roofline_conv": c5.get("roofline_conv")}
            except (subprocess.TimeoutExpired, IndexError, ValueError, KeyError) as e:
                out["config5"] = {"value": None, "note": f"no line within {args.config5_timeout}s ({type(e).__name__}): {stderr_tail}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
