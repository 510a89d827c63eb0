"""GPU: the multi-rank fine-tuning path end to end on ONE GPU -- two processes share cuda:0 and exchange the
flat gradient buffer (+ loss slot) through torch.distributed (gloo transport here, because RCCL refuses two
ranks on one device; on a multi-GPU node the same code runs over RCCL/xGMI, one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

WORKER = r"""
import argparse, json, os, sys, torch
sys.path.insert(0, %(repo)r)
from consistent_depth_amd import parallel, synthetic
from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
rank, local_rank, world = parallel.init()
dev = parallel.local_device(local_rank); torch.cuda.set_device(dev)
params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4, optimizer="Adam")
model = get_depth_model("mc")(seed=0); model.train()
eager = FineTuneStep(model, params, world=world)
GRAPH = %(graph)d
step = GraphedFineTuneStep(eager, eager_steps=1) if GRAPH else eager
t = lambda a: torch.tensor(a, device=dev)
losses = []
for it in range(%(iters)d):
    b = synthetic.make_scene_batch(2, 64, 48, seed=10 * it + rank)          # every rank trains on its own pairs
    imgs = torch.rand(2, 2, 3, 64, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(100 * it + rank))
    meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
            "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
    loss, _ = step(imgs, meta)
    losses.append(loss.item())
w = eager.opt.flat_param
ws = [torch.zeros_like(w) for _ in range(world)]
torch.distributed.all_gather(ws, w)
same = all(torch.equal(ws[0], x) for x in ws)
global_batch = None
if rank == 0:
    # The same steps as ONE process that owns the global batch: the two ranks' shards evaluated one after the other on the same
    # weights (BatchNorm statistics stay per shard, like on the ranks), gradients and losses SUMMED, one guarded Adam step with
    # the 1/world factor -- through the same FineTuneStep pieces.  fp32 addition of two buffers is commutative, so the weights
    # must come out bit for bit.
    model1 = get_depth_model("mc")(seed=0); model1.train()
    one = FineTuneStep(model1, params, world=1)
    for it in range(%(iters)d):
        gsum, lsum = torch.zeros_like(one.opt.flat_grad), torch.zeros(1, device=dev)
        for r in range(world):
            b = synthetic.make_scene_batch(2, 64, 48, seed=10 * it + r)
            imgs = torch.rand(2, 2, 3, 64, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(100 * it + r))
            meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
                    "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
            guard, _ = one._grads(imgs, meta)
            gsum += one.opt.flat_grad
            lsum += guard.reshape(1)
        one.opt.flat_grad.copy_(gsum)
        one.opt.step(grad_scale=1.0 / world, guard_loss=lsum)
    d = (one.opt.flat_param - w).abs()
    global_batch = {"bitwise": bool(torch.equal(one.opt.flat_param, w)), "max_abs": float(d.max())}
if rank == 0:
    print("RESULT " + json.dumps({"world": world, "same_weights": same, "losses": losses, "steps": int(eager.opt.step_dev.item()),
                                  "graphed": getattr(step, "graphed", None), "capture_error": getattr(step, "capture_error", None),
                                  "global_batch": global_batch}))
torch.distributed.barrier(); torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("graph", [0, 1], ids=["eager", "hip_graph"])
def test_two_ranks_share_one_gpu_and_stay_in_sync(tmp_path, graph):
    iters = 4 if graph else 2      # graph: 1 eager step, capture + 3 replays; the all-reduce and Adam stay outside the graph
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "graph": graph, "iters": iters})
    env = dict(os.environ, CD_AMD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29541 + graph), str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(lines[-1][len("RESULT "):])
    assert res["world"] == 2 and res["steps"] == iters
    if graph:
        assert res["graphed"] is True, res["capture_error"]
    assert res["same_weights"], "ranks diverged: the gradient all-reduce / guarded Adam are not in lock-step"
    assert all(l == l for l in res["losses"])
    # 2 ranks x B pairs == 1 process with the 2B global batch (shards evaluated in turn), through FineTuneStep: same weights
    gb = res["global_batch"]
    assert gb["max_abs"] <= 1e-7, gb
    print("2 ranks vs one process with the global batch:", gb)


NCCL_WORKER = r"""
import argparse, json, os, sys, torch
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from consistent_depth_amd import parallel, synthetic
from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # the branch parallel.init takes on a multi-GPU node
params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4, optimizer="Adam")
model = get_depth_model("mc")(seed=0); model.train()
eager = FineTuneStep(model, params, world=1)
step = GraphedFineTuneStep(eager, eager_steps=1)
t = lambda a: torch.tensor(a, device=dev)
b = synthetic.make_scene_batch(2, 64, 48, seed=1)
imgs = torch.rand(2, 2, 3, 64, 48, device=dev)
meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
        "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}
# the data-parallel exchange on the REAL buffer: [flat grads | loss slot], one RCCL call, in place
guard, _ = eager._grads(imgs, meta)
eager.opt.loss_slot.copy_(guard.reshape(1))
before = eager.opt.reduce_buffer.clone()
dist.all_reduce(eager.opt.reduce_buffer, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
same = torch.equal(before, eager.opt.reduce_buffer)       # world 1: the sum is the buffer itself
eager._update(eager.opt.loss_slot)
losses = [step(imgs, meta)[0].item() for _ in range(4)]   # graph capture with RCCL's watchdog thread alive
print("RESULT " + json.dumps({"backend": dist.get_backend(), "allreduce_identity": same, "losses": losses,
                              "graphed": step.graphed, "capture_error": step.capture_error, "bytes": eager.opt.reduce_buffer.numel() * 4}))
dist.barrier(); dist.destroy_process_group()
"""


def test_rccl_one_rank_smoke(tmp_path):
    """RCCL itself (torch.distributed backend "nccl" on ROCm) on this box's single GPU: process-group init with device_id, the
    flat gradient all-reduce on the optimiser's real buffer, then eager + graph-replayed steps with the communicator alive.
    (Two ranks cannot share a device under RCCL; the 2-rank logic above runs over gloo.)"""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % {"repo": REPO})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(lines[-1][len("RESULT "):])
    assert res["backend"] == "nccl" and res["allreduce_identity"] and res["bytes"] > 20e6
    assert res["graphed"] is True, res["capture_error"]
    assert all(l == l for l in res["losses"])


def test_c_abi_allreduce_on_a_raw_rccl_communicator():
    """cd_allreduce_mean_f32: the collective of the data-parallel step straight from the C ABI, on a communicator created
    with RCCL's own C API (ncclGetUniqueId / ncclCommInitRank, world = 1 on this box) -- no torch.distributed involved."""
    import ctypes
    import torch
    from consistent_depth_amd import _native
    lib = _native.lib()
    assert lib.cd_rccl_available() == 1
    rccl = None
    for name in ("librccl.so", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = ctypes.CDLL(name)
            break
        except OSError:
            continue
    assert rccl is not None

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid, comm = UniqueId(), ctypes.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    buf = torch.randn(5_357_731, device="cuda")          # the hourglass gradient + the loss slot
    want = buf.clone()
    assert lib.cd_allreduce_mean_f32(buf.data_ptr(), buf.numel(), comm, 1, _native.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(buf, want)                        # world 1: sum / 1
    # the scale path (what world = 4 would do to the summed buffer), without a communicator
    assert lib.cd_allreduce_mean_f32(buf.data_ptr(), buf.numel(), None, 1, _native.stream_ptr()) == 0
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)


COLLECTIVE_GRAPH_WORKER = r"""
import argparse, json, os, sys
import torch
sys.path.insert(0, %(repo)r)
import torch.distributed as dist
from consistent_depth_amd import synthetic
from consistent_depth_amd.engine import FineTuneStep, GraphedFineTuneStep
from consistent_depth_amd.monodepth.depth_model_registry import get_depth_model
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
params = argparse.Namespace(lambda_reprojection=1.0, lambda_view_baseline=0.1, lambda_parameter=0, learning_rate=4e-4, optimizer="Adam")
t = lambda a: torch.tensor(a, device=dev)
b = synthetic.make_scene_batch(2, 64, 48, seed=1)
imgs = torch.rand(2, 2, 3, 64, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
meta = {"intrinsics": t(b["intrinsics"]), "extrinsics": t(b["extrinsics"]),
        "geometry_consistency": {"flows": [t(f) for f in b["flows"]], "masks": [t(m) for m in b["masks"]]}}

def run(graph_collective):
    os.environ["CD_AMD_DP_GRAPH_COLLECTIVE"] = "1" if graph_collective else "0"
    model = get_depth_model("mc")(seed=0); model.train()
    # world = 2 on a ONE-rank communicator: the step takes its multi-rank branch (loss slot, RCCL all-reduce of [grads | loss],
    # 1/world in Adam); the sum over the one rank is the buffer itself
    step = GraphedFineTuneStep(FineTuneStep(model, params, world=2), eager_steps=1)
    losses = [step(imgs, meta)[0].item() for _ in range(5)]
    torch.cuda.synchronize()
    return losses, step.graphed, step.graph_collective, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()

l0, g0, c0, p0 = run(False)
l1, g1, c1, p1 = run(True)
print("RESULT " + json.dumps({"losses_eager_collective": l0, "losses_graphed_collective": l1, "graphed": [g0, g1], "collective_in_graph": [c0, c1],
                              "params_equal": bool(torch.equal(p0, p1))}))
dist.barrier(); dist.destroy_process_group()
"""


def test_rccl_collective_inside_the_step_graph(tmp_path):
    """CD_AMD_DP_GRAPH_COLLECTIVE=1: the gradient all-reduce (RCCL, backend "nccl") and the Adam launch are captured INTO the step's HIP
    graph, so a data-parallel step is one replay per rank.  On this one-GPU box the communicator has one rank while the step runs
    its multi-rank branch (world = 2 in FineTuneStep): the trajectory must equal, bit for bit, the default (collective + Adam launched
    eagerly after the replay)."""
    script = tmp_path / "collective_graph_worker.py"
    script.write_text(COLLECTIVE_GRAPH_WORKER % {"repo": REPO})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(lines[-1][len("RESULT "):])
    assert res["graphed"] == [True, True] and res["collective_in_graph"] == [False, True], res
    assert res["losses_eager_collective"] == res["losses_graphed_collective"] and res["params_equal"], res


def test_bench_main_two_ranks_end_to_end(tmp_path):
    """bench.py's own main() with --gpus 2, end to end on this box's one GPU (gloo transport between the two ranks, because RCCL
    refuses two ranks on one device): the self-launch under torch.distributed.run, the shared-seed shards of EpochPlans, graph
    replay + eager all-reduce + Adam, barrier-bracketed timing with the maximum over ranks, the finite-loss decision taken on the
    minimum over ranks, rank 0 printing ONE line -- the path the driver's multi-GPU tier runs (over RCCL)."""
    env = dict(os.environ, CD_AMD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--frames", "10",
                        "--height", "64", "--width", "48", "--no-loss-microbench", "--no-cpu-baseline", "--no-config5"],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    cfg = out["config"]
    assert cfg["global_batch"] == 8 and cfg["parallelism"] == "dp2" and cfg["finite_loss_steps"] == 3 and cfg["hip_graph"] is True
    assert "DP over 2 GPUs" in cfg["workload"] and "configs[3]:" not in cfg["workload"]      # a 10-frame clip is not configs[3]
    assert "eager, after the graph replay" in cfg["dp_exchange"]
    assert out["value"] == pytest.approx(8 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-3)
