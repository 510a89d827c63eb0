"""CPU (gloo, world_size 2): the data-parallel plumbing of consistent_depth_amd.parallel --
pair-list sharding and the single flat all-reduce (gradients + loss slot) with the 1/world scale."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from consistent_depth_amd import parallel


def test_shard_indices_cover_and_balance():
    n, bs = 715, 4  # BASELINE C1: 244 frames hierarchical2 -> 715 pairs
    for world in (1, 2, 4, 8):
        plans = [parallel.shard_indices(n, epoch=3, seed=0, rank=r, world=world, batch_size=bs) for r in range(world)]
        steps = {len(p) for p in plans}
        assert len(steps) == 1, "all ranks must run the same number of steps"
        seen = [i for p in plans for step in p for i in step]
        assert len(seen) == len(set(seen)), "no pair may be trained twice in an epoch"
        if world == 1:
            assert sorted(seen) == list(range(n)) and len(plans[0]) == 179  # 179 it/epoch, last batch short
            assert len(plans[0][-1]) == n - 178 * bs
        else:
            assert len(seen) >= n - bs * world
        for s in range(len(plans[0])):
            sizes = {len(p[s]) for p in plans}
            assert len(sizes) == 1
    # same (seed, epoch) -> same permutation on every rank; different epochs differ
    a = parallel.shard_indices(n, 0, 0, 0, 1, bs)
    b = parallel.shard_indices(n, 1, 0, 0, 1, bs)
    assert a != b and a == parallel.shard_indices(n, 0, 0, 0, 1, bs)
    # validation order: unshuffled
    v = parallel.shard_indices(10, 0, 0, 0, 1, 4, shuffle=False)
    assert v == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    torch.manual_seed(0)
    # a tiny surrogate model: loss_b = mean over the rank's pairs of (w . x_b)^2
    wvec = torch.nn.Parameter(torch.linspace(-1, 1, 37))
    data = torch.randn(8, 37, generator=torch.Generator().manual_seed(5))
    plan = parallel.shard_indices(8, 0, 0, rank, world, batch_size=4 // world * world // world if False else 4 // world)
    ids = plan[0]
    loss = ((data[ids] @ wvec) ** 2).mean()
    loss.backward()
    # flat reduce buffer = [grads | pad | loss slot], like FlatAdam.reduce_buffer
    buf = torch.zeros(64 + 64)
    buf[:37] = wvec.grad
    buf[64] = loss.detach()
    parallel.allreduce_sum_(buf)
    grad_mean = buf[:37] / world          # 1/world is folded into the Adam kernel (grad_scale)
    loss_sum = buf[64]
    # reference: one process with the global batch
    w2 = torch.nn.Parameter(torch.linspace(-1, 1, 37))
    all_ids = [i for rr in range(world) for i in parallel.shard_indices(8, 0, 0, rr, world, 4 // world)[0]]
    ref_loss = ((data[all_ids] @ w2) ** 2).mean()
    ref_loss.backward()
    ok = torch.allclose(grad_mean, w2.grad, rtol=1e-5, atol=1e-6) and abs(loss_sum.item() / world - ref_loss.item()) < 1e-5
    # NaN on one rank poisons the summed guard scalar on every rank -> all ranks skip together
    g = torch.tensor([float("nan") if rank == 1 else 1.0])
    parallel.allreduce_sum_(g)
    ok = ok and bool(torch.isnan(g).item())
    t = [torch.full((3,), float(rank))]
    parallel.broadcast_(t, src=0)
    ok = ok and bool((t[0] == 0).all())
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_eval_shards_cover_every_pair_exactly_once():
    """Validation evaluates EVERY pair (the reference's sweep does): 11 pairs on 2 ranks with batch 4 used to lose pair 10,
    7 pairs on 8 ranks used to evaluate nothing."""
    for n, world, bs in ((11, 2, 4), (7, 8, 4), (715, 8, 4), (1, 3, 4), (0, 2, 4)):
        seen = []
        for r in range(world):
            for ids in parallel.eval_shard(n, r, world, bs):
                assert 0 < len(ids) <= bs
                seen += ids
        assert sorted(seen) == list(range(n))


def test_plan_to_device_roundtrip():
    import torch
    plan = [[3, 1, 2, 0], [5, 4]]
    dev = parallel.plan_to_device(plan, torch.device("cpu"))
    assert [d.tolist() for d in dev] == plan and all(d.dtype == torch.int64 and d.is_contiguous() for d in dev)
    assert parallel.plan_to_device([], torch.device("cpu")) == []


def test_eval_chunks_are_the_single_process_batches_and_frames_have_one_owner():
    """Multi-rank validation = the reference's sequential sweep dealt out batch by batch: the union over ranks is exactly the
    1-rank batch list (so train-mode BatchNorm sees the same batches whatever the world size), and every frame's export
    belongs to exactly one (rank, batch) -- the batch of its first sighting in sweep order (depth_fine_tuning.py:343-360)."""
    pairs = [[0, 1], [2, 4], [1, 2], [0, 4], [3, 4], [2, 3], [0, 2], [4, 5], [1, 3], [3, 5], [5, 6]]
    bs = 4
    single = parallel.eval_chunks(len(pairs), 0, 1, bs)
    assert [ids for _, ids in single] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]
    first = parallel.first_sightings(pairs, bs)
    assert first == {0: 0, 1: 0, 2: 0, 4: 0, 3: 1, 5: 1, 6: 2}
    for world in (2, 3, 8):
        dealt = sorted((c, ids) for r in range(world) for c, ids in parallel.eval_chunks(len(pairs), r, world, bs))
        assert dealt == single
        owners = {}
        for r in range(world):
            for c, ids in parallel.eval_chunks(len(pairs), r, world, bs):
                for pid in ids:
                    for f in pairs[pid]:
                        if first[f] == c:
                            owners.setdefault(f, set()).add(r)
        assert set(owners) == set(first) and all(len(v) == 1 for v in owners.values())


class _FlatStandIn:
    """The part of optimizer.FlatAdam that parallel.GradBuckets uses (flat_grad, loss_slot, _params, _offsets), on the CPU."""

    def __init__(self, params, align=4):
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self._store = torch.zeros(n + align)
        self.flat_grad, self.loss_slot = self._store[:n], self._store[n:n + 1]
        self._params, self._offsets = list(params), offs
        for p, o in zip(params, offs):
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    unused = torch.nn.Parameter(torch.zeros(5))                 # never receives a gradient (like the hourglass' confidence head)
    params = [unused] + list(net.parameters())                   # (in the first bucket: the one that completes last anyway)
    opt = _FlatStandIn(params)
    buckets = parallel.GradBuckets(opt, n_buckets=3)
    assert len(buckets.ranges) == 3 and buckets.ranges[0][0] == 0 and buckets.ranges[-1][1] == opt.flat_grad.numel()
    assert all(a[1] == b[0] for a, b in zip(buckets.ranges, buckets.ranges[1:]))
    data = torch.randn(8, 12, generator=torch.Generator().manual_seed(3))
    launches = []
    orig = buckets._launch
    buckets._launch = lambda b: (launches.append(b), orig(b))[1]
    ok = True
    for it in range(2):                                          # two steps: the buckets re-arm
        ids = list(range(rank * 4, rank * 4 + 4))
        opt._store.zero_()
        loss = net(data[ids]).pow(2).mean()
        opt.loss_slot.copy_(loss.detach().reshape(1))
        buckets.arm()
        loss.backward()
        n_during = len(launches)                                 # buckets of the LAST layers were reduced while backward ran
        buckets.finish()
        ref = torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
        ref.load_state_dict(net.state_dict())
        ref_loss = ref(data).pow(2).mean()                       # the global batch in one process
        ref_loss.backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            ok = ok and torch.allclose(p.grad / world, q.grad, rtol=1e-5, atol=1e-7)
        ok = ok and abs(opt.loss_slot.item() / world - ref_loss.item()) < 1e-6 and bool((unused.grad == 0).all())
        ok = ok and n_during >= 2 + 3 * it and sorted(launches[3 * it:]) == [0, 1, 2]     # buckets 2 and 1 went out during backward
        ok = ok and launches[3 * it] == 2                        # reverse layer order: the last bucket goes first
    buckets.close()
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_overlaps_backward_and_matches_the_global_batch():
    """parallel.GradBuckets (the MiDaS-sized gradient: 420 MB in 4 buckets): ranges tile the flat buffer, buckets go out in
    reverse layer order WHILE autograd is still running, parameters without a gradient do not block their bucket, and the reduced
    gradient / loss equal the single-process global batch."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _one_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = []
    orig = dist.all_reduce

    def counting(t, *a, **kw):
        calls.append(t.numel())
        return orig(t, *a, **kw)
    dist.all_reduce = counting
    try:
        assert parallel.collective_active()
        buf = torch.arange(5.0)
        parallel.allreduce_sum_(buf)                                   # the flat path
        net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
        opt = _FlatStandIn(list(net.parameters()))
        buckets = parallel.GradBuckets(opt, n_buckets=2)
        loss = net(torch.randn(4, 6)).pow(2).mean()
        opt.loss_slot.copy_(loss.detach().reshape(1))
        before = opt.loss_slot.clone()
        buckets.arm()
        loss.backward()
        buckets.finish()                                               # the bucketed path: 2 buckets + the loss slot
        buckets.close()
        hooks_gone = buckets._handles == []
    finally:
        dist.all_reduce = orig
    out[0] = (calls[0] == 5 and len(calls) == 4 and sorted(calls[1:])[0] == 1 and torch.equal(buf, torch.arange(5.0))
              and torch.equal(before, opt.loss_slot) and hooks_gone)
    dist.destroy_process_group()


def test_one_rank_group_runs_the_collectives_of_both_exchange_paths():
    """ONE predicate (parallel.collective_active): with a process group of a single rank the flat all-reduce AND the bucketed one
    (gradient buckets + loss slot) issue their collectives -- that is how the RCCL launches of a captured step are exercised on a
    one-GPU box; round 3's bucketed path skipped them for world == 1 while the flat path did not."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_rank_worker, args=(1, port, out), nprocs=1, join=True)
    assert dict(out) == {0: True}


def test_bench_epoch_plans_deal_disjoint_full_batches_to_the_ranks():
    """bench.py's schedule (EpochPlans over parallel.shard_indices): at every step the ranks hold disjoint pairs of ONE shared
    permutation, all batches are full (one graph signature), and every rank sees the same number of steps per epoch -- so the ranks
    issue the same sequence of collectives."""
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_plans", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n_pairs, world, bs = 715, 4, 4
    plans = [bench.EpochPlans(n_pairs, r, world, bs, torch.device("cpu"), seed=0) for r in range(world)]
    seen_epoch0 = []
    for step in range(2 * (n_pairs // (world * bs)) + 3):          # runs into the third epoch
        batches = [p.next() for p in plans]
        assert all(len(b) == bs for b in batches)
        assert len({p.epoch for p in plans}) == 1 and len({p.pos for p in plans}) == 1       # same epoch, same step on every rank
        flat = torch.cat([torch.as_tensor(b).reshape(-1) for b in batches]).tolist()
        assert len(set(flat)) == world * bs and all(0 <= i < n_pairs for i in flat)
        if plans[0].epoch == 0:
            seen_epoch0 += flat
    assert len(set(seen_epoch0)) == len(seen_epoch0) == (n_pairs // (world * bs)) * world * bs    # no pair twice within an epoch


class _FakeGraph:
    def __init__(self, with_collective):
        self.with_collective, self.replays = with_collective, 0

    def replay(self):
        self.replays += 1
        if self.with_collective:          # a graph captured WITH the exchange issues the collective when replayed
            parallel.allreduce_sum_(torch.ones(4))


class _FakeStep:
    """The surface GraphedFineTuneStep uses of a FineTuneStep, with the data-parallel exchange as its only real action."""
    world = 2

    def __init__(self):
        self.eager, self.updates = 0, 0

    def __call__(self, images, metadata):
        self.eager += 1
        parallel.allreduce_sum_(torch.ones(4))
        return torch.zeros(1), {}

    def _update(self, guard):
        self.updates += 1
        parallel.allreduce_sum_(torch.ones(4))

    def _weights_updated(self):
        pass


def _consensus_worker(rank, world, port, fail_rank, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CD_AMD_DP_GRAPH_COLLECTIVE="1")
    parallel.init(backend="gloo")
    from consistent_depth_amd.engine import GraphedFineTuneStep
    step = _FakeStep()
    g = GraphedFineTuneStep(step, eager_steps=1)
    assert g.graph_collective
    captures = []

    def fake_capture(images, metadata):
        captures.append(g.graph_collective)
        if g.graph_collective and rank == fail_rank:
            raise RuntimeError("forced capture failure")
        return {"graph": _FakeGraph(g.graph_collective), "images": images.clone(), "meta": {}, "flat": [], "guard": torch.zeros(1), "parts": {}}
    g._capture = fake_capture
    x = torch.zeros(2, 2, 3, 4, 4)
    for _ in range(4):          # 1 eager call, the capture call, 2 replays: every rank issues the same collectives in the same order
        g(x, {})
    graph = next(iter(g._graphs.values()))["graph"]
    out[rank] = dict(collective=g.graph_collective, graphed=g.graphed, eager=step.eager, updates=step.updates, captures=captures,
                     replays=graph.replays, with_collective=graph.with_collective, error=g.capture_error)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [1, -1], ids=["one_rank_fails", "all_capture"])
def test_capture_consensus_of_the_graphed_exchange(fail_rank):
    """CD_AMD_DP_GRAPH_COLLECTIVE=1 (the gradient all-reduce inside the step graph): after the capture attempt the ranks agree whether
    EVERYBODY holds a graph (parallel.all_agree).  One rank forced to fail -> all ranks drop to the eager exchange together (graphs
    without the collective + an eager all-reduce per step) and keep issuing their collectives in the same order: the run finishes
    instead of hanging; nobody fails -> the collective stays inside the replayed graph on every rank."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_consensus_worker, args=(world, port, fail_rank, out), nprocs=world, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    for r, o in res.items():
        assert o["graphed"] is True and o["eager"] == 1 and o["replays"] == 3
        if fail_rank < 0:
            assert o["collective"] is True and o["with_collective"] is True and o["updates"] == 0 and o["captures"] == [True]
        else:
            assert o["collective"] is False and o["with_collective"] is False and o["updates"] == 3 and o["captures"] == [True, False]
            assert "eager exchange on every rank" in o["error"]
