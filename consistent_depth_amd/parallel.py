"""Data parallelism over frame pairs: one process per GPU, gradients summed with ONE RCCL
all-reduce of the flat gradient buffer over xGMI.

The reference's only multi-GPU mechanism is single-process nn.DataParallel
(/root/reference/monodepth/midas_v2_model.py:41-43) plus "batch_size *= num_gpus"
(depth_fine_tuning.py:155-159): scatter the batch, replicate the module each forward, gather
outputs, reduce-add grads on GPU 0.  Here every rank owns a replica and a shard of the pair list
(frame pairs are independent units); per-rank loss = mean over its pairs,
global objective = mean over ranks, so gradients are all-reduced (sum) and the 1/world factor
is folded into the Adam kernel.  BatchNorm batch statistics stay rank-local (8 images per
rank), which is what DataParallel does too.

torch.distributed with backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests of
the sharding logic.  The payload for `mc` is 21.4 MB fp32 per step: on MI355X's 7 x ~153 GB/s
point-to-point xGMI links that is tens of microseconds with a direct (all-links) algorithm and
~0.25 ms with a ring, against a >=30 ms step -- a single un-bucketed collective is the right
granularity here; bucketing only pays for the ~420 MB MiDaS gradient.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str = None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # CD_AMD_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses duplicate devices): used by the
        # single-GPU test of the multi-rank path; production is "nccl" (= RCCL on ROCm), one GPU per rank
        backend = backend or os.environ.get("CD_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>, wrapped onto the visible devices (several ranks may share a GPU in tests)."""
    return torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def collective_active() -> bool:
    """ONE predicate for every data-parallel exchange (flat all-reduce, bucketed all-reduce, loss slot): a process group exists.
    The callers decide with their own `world` whether there is anything to exchange; with a one-rank group the collective still
    runs -- that is how the RCCL launch inside a captured step, flat or bucketed, is exercised on a one-GPU box
    (tests/test_dp_gpu.py)."""
    return dist.is_initialized()


class ExchangeTimer:
    """Optional HIP-event bracket around the step's gradient all-reduce (bench.py: `dp_exchange_ms`), so that the first run on a
    multi-GPU node says by itself how long the collective takes per step.  Events are recorded on the current stream (the one the
    collective is enqueued on: ProcessGroupNCCL orders its own stream after it and the current stream after its work) -- the
    elapsed time is enqueue-to-completion of the exchange as the step sees it.  Off unless `start(n)` was called."""

    def __init__(self):
        self.events, self.pos = [], 0

    def start(self, n):
        self.events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        self.pos = 0

    def stop(self):
        """-> milliseconds of every bracket recorded since start() (call after a device synchronisation)."""
        ms = [a.elapsed_time(b) for a, b in self.events[:self.pos]]
        self.events, self.pos = [], 0
        return ms

    def bracket(self):
        if self.pos < len(self.events):
            self.pos += 1
            return self.events[self.pos - 1]
        return None


exchange_timer = ExchangeTimer()


def allreduce_sum_(buf: torch.Tensor):
    """In-place sum over ranks of one flat fp32 buffer (no-op without a process group; see collective_active)."""
    if collective_active():
        ev = exchange_timer.bracket()
        if ev is not None:
            ev[0].record()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if ev is not None:
            ev[1].record()
    return buf


class GradBuckets:
    """Bucketed gradient all-reduce OVERLAPPED with the backward pass, for models whose backward is driven by autograd (the
    MiDaS-shaped backbone: ~105 M parameters = 420 MB of fp32 gradients; the reference's counterpart is nn.DataParallel's
    reduce-add onto GPU 0, monodepth/midas_v2_model.py:41-43).

    The optimiser's flat gradient buffer (optimizer.FlatAdam: parameters in registration order, each a view) is cut into
    `n_buckets` contiguous ranges of about equal size.  A post-accumulate hook on every parameter counts its bucket down; when
    the LAST gradient of a bucket has been written the bucket's slice is all-reduced asynchronously (RCCL runs it on its own
    stream, ordered after the kernels that wrote the slice), while autograd keeps producing the gradients of earlier layers.
    `finish()` reduces the loss slot (the NaN guard must see the summed loss) and makes the current stream wait for every bucket.
    Buckets are ranges of ONE buffer: no packing copies, and Adam stays one launch over the flat buffer.

    Sizing for MI355X: xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 420 MB ring all-reduce over 8 GPUs moves
    2 x 7/8 x 420 MB per GPU = 735 MB through the links -- ~0.7 ms when the algorithm uses all links, ~5 ms as a single ring --
    against a backward pass of ~100 ms.  A handful of large buckets (default 4, ~105 MB each) keeps every collective far above
    the latency-bound regime and still hides all but the last one (the stem's, ~1/4 of the total) behind the backward pass.

    The hourglass engine (`mc`) writes its gradients itself, without autograd hooks, and its payload is 21 MB: it keeps the
    single un-bucketed collective of FineTuneStep.
    """

    def __init__(self, opt, n_buckets: int = 4):
        self.opt = opt
        params, offs = opt._params, opt._offsets
        total = opt.flat_grad.numel()
        n_buckets = max(1, min(n_buckets, len(params)))
        target = (total + n_buckets - 1) // n_buckets
        self.ranges, self._bucket_of, lo, b = [], {}, 0, 0     # bucket b = flat_grad[lo:hi]
        counts = []
        for i, (p, o) in enumerate(zip(params, offs)):
            end = offs[i + 1] if i + 1 < len(params) else total
            self._bucket_of[id(p)] = b
            if len(counts) <= b:
                counts.append(0)
            counts[b] += 1
            if end - lo >= target or i + 1 == len(params):
                self.ranges.append((lo, end))
                lo, b = end, b + 1
        self._counts = counts
        self._pending, self._work = list(counts), []
        self._armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]

    def arm(self):
        """Call before backward(): every parameter with requires_grad is expected to receive a gradient (parameters that
        receive none -- e.g. an unused head -- leave their bucket to `finish`)."""
        self._pending, self._work, self._armed = list(self._counts), [], True
        self._launched = [False] * len(self.ranges)

    def _launch(self, b):
        lo, hi = self.ranges[b]
        self._launched[b] = True
        if collective_active():
            self._work.append(dist.all_reduce(self.opt.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def finish(self):
        """After backward(): launch what is left (buckets holding a parameter that got no gradient), reduce the loss slot, wait."""
        for b in range(len(self.ranges)):
            if not self._launched[b]:
                self._launch(b)
        if collective_active():
            self._work.append(dist.all_reduce(self.opt.loss_slot, op=dist.ReduceOp.SUM, async_op=True))
        for w in self._work:
            w.wait()
        self._work, self._armed = [], False

    def close(self):
        """Remove the parameter hooks (FineTuneStep.close() calls this when the step is discarded)."""
        for h in self._handles:
            h.remove()
        self._handles = []


def all_agree(ok: bool, device=None) -> bool:
    """True iff `ok` holds on EVERY rank (one tiny all-reduce(MIN); every rank must call it at the same point of its call sequence).
    Used as the capture-consensus handshake of GraphedFineTuneStep: a step graph that contains the gradient all-reduce may only be
    replayed if every rank captured one -- a rank that fell back to eager steps would issue its collectives in another order."""
    if not collective_active():
        return bool(ok)
    backend = dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def broadcast_(tensors, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def eval_shard(n_items: int, rank: int, world: int, batch_size: int):
    """Validation shards: the unshuffled pair list is cut into the reference's sequential batches of `batch_size`
    (depth_fine_tuning.py:215-218, the last one short) and rank r takes batches r, r + world, ... -- EVERY batch has the
    pairs it has in a single-process sweep, so the train-mode BatchNorm statistics, hence every per-pair loss and every
    exported depth map, do not depend on the number of ranks (nn.DataParallel's scatter cuts the same contiguous chunks).
    Ranks may run different numbers of batches: there is no collective inside the validation loop."""
    return [ids for _, ids in eval_chunks(n_items, rank, world, batch_size)]


def eval_chunks(n_items: int, rank: int, world: int, batch_size: int):
    """[(global batch number, pair ids)] of this rank's share of the validation sweep (see eval_shard)."""
    chunks = [list(range(s, min(s + batch_size, n_items))) for s in range(0, n_items, batch_size)]
    return [(c, ids) for c, ids in enumerate(chunks) if c % world == rank]


def first_sightings(frames_of_pair, batch_size: int):
    """{frame: global batch number of the sweep in which the frame is seen first}.  The reference exports a frame's
    validation depth at its FIRST sighting in the sequential sweep (depth_fine_tuning.py:343-360); with the sweep sharded
    over ranks, exactly the rank that owns that batch writes the file -- decided on the host from the unshuffled plan,
    identically on every rank, no communication."""
    first = {}
    for pid, frames in enumerate(frames_of_pair):
        for f in frames:
            first.setdefault(f, pid // batch_size)
    return first


def plan_to_device(plan, device):
    """The per-step index lists of an epoch as int64 device tensors (ONE upload; slices of a padded matrix)."""
    if not plan:
        return []
    import numpy as np
    width = max(len(ids) for ids in plan)
    mat = np.zeros((len(plan), width), np.int64)
    for r, ids in enumerate(plan):
        mat[r, :len(ids)] = ids
    dev = torch.as_tensor(mat).to(device)
    return [dev[r, :len(ids)].contiguous() for r, ids in enumerate(plan)]


def shard_indices(n_items: int, epoch: int, seed: int, rank: int, world: int, batch_size: int, shuffle: bool = True):
    """Pair indices this rank trains on in `epoch`.

    A permutation shared by all ranks (numpy PCG64 seeded by (seed, epoch)) is cut into global
    batches of batch_size*world; inside every global batch rank r takes the r-th slice of
    batch_size.  With world == 1 this is exactly DataLoader(shuffle=True, drop_last=False)
    batching (the last batch may be short); with world > 1 the tail that cannot give every rank
    at least one pair is dropped so all ranks run the same number of steps.
    Returns a list of per-step index lists.
    """
    import numpy as np
    order = np.random.default_rng([seed, epoch]).permutation(n_items) if shuffle else np.arange(n_items)
    steps = []
    gb = batch_size * world
    for start in range(0, n_items, gb):
        chunk = order[start:start + gb]
        if world == 1:
            steps.append(chunk.tolist())
            continue
        if len(chunk) < world:
            break
        per = len(chunk) // world
        per = min(per, batch_size)
        steps.append(chunk[rank * per:(rank + 1) * per].tolist())
    return steps
