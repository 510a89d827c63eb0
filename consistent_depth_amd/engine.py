"""One fine-tuning step of the hot path (reference loop body: depth_fine_tuning.py:264-293).

    raw   = model.estimate_raw(images)          # CNN forward (hourglass / MiDaS)
    loss  = JointLoss(raw, metadata)            # fused HIP loss, depth head fused in
    loss.backward()                             # CNN backward
    [all-reduce flat grads + loss scalar]       # data parallel only, one RCCL call
    FlatAdam.step(guard_loss=loss)              # one HIP launch; NaN guard on the device

Compared with the reference's body there is no per-step host synchronisation: the reference
prints `loss[0]` (a D2H sync) and tests `torch.isnan(loss)` on the host every step (:275-280);
here the NaN-skip is evaluated inside the Adam kernel and losses are fetched only when logged.
"""
from __future__ import annotations

import os

import torch

from . import optimizer as cd_optimizer
from . import parallel
from .loss.joint_loss import JointLoss


class FineTuneStep:
    def __init__(self, model, params, world: int = None):
        """`params` needs lambda_reprojection / lambda_view_baseline / lambda_parameter /
        learning_rate / optimizer (as produced by consistent_depth_amd.params)."""
        self.model = model
        self.world = world if world is not None else parallel.world_size()
        plist = list(model.parameters())
        init = [p.detach().clone() for p in plist] if params.lambda_parameter > 0 else None
        self.criterion = JointLoss(params, parameters_init=init, depth_mode=model.depth_mode)
        self.opt = cd_optimizer.create(getattr(params, "optimizer", "Adam"), plist, params.learning_rate,
                                       betas=(0.9, 0.999))
        self._plist = plist
        self._root_grad = None       # d loss / d loss = 1, kept (loss.backward() would fill a fresh ones_like every step)
        # Autograd-driven models with a large gradient (MiDaS: 420 MB): bucketed all-reduce overlapped with the backward pass.
        # The hourglass engine writes its 21 MB of gradients without autograd hooks: one collective after the backward.
        self.buckets = None
        n_buckets = int(os.environ.get("CD_AMD_DP_BUCKETS", "4"))
        if self.world > 1 and n_buckets > 1 and getattr(model, "_engine", None) is None and self.opt.flat_grad.numel() >= (8 << 20):
            self.buckets = parallel.GradBuckets(self.opt, n_buckets)

    def close(self):
        """Release what the step hooked into the model (the bucketed all-reduce's parameter hooks); call when a step object is
        replaced by another one on the same model."""
        if self.buckets is not None:
            self.buckets.close()
            self.buckets = None

    def _backward(self, loss):
        if self._root_grad is None or self._root_grad.shape != loss.shape or self._root_grad.device != loss.device:
            self._root_grad = torch.ones_like(loss)
        torch.autograd.backward(loss, grad_tensors=self._root_grad)

    def _backward_and_reduce(self, loss):
        """backward + the data-parallel exchange; returns the guard scalar (the summed loss with world > 1)."""
        guard = loss.detach()
        if self.world == 1:
            self._backward(loss)
            return guard
        if self.buckets is not None:
            self.opt.loss_slot.copy_(guard.reshape(1))     # before backward: nothing writes the slot afterwards
            self.buckets.arm()
            self._backward(loss)
            self.buckets.finish()
            return self.opt.loss_slot
        self._backward(loss)
        self.opt.loss_slot.copy_(guard.reshape(1))
        parallel.allreduce_sum_(self.opt.reduce_buffer)
        return self.opt.loss_slot

    def _weights_updated(self):
        """Tell the model that the optimiser moved its weights (it writes through raw pointers, so tensor version counters do
        not change): derived copies -- the MiDaS backbone's packed filters (ops/conv_layer.py::PackPool) -- are stale."""
        hook = getattr(self.model, "weights_updated", None)
        if hook is not None:
            hook()

    def forward_loss(self, images, metadata):
        raw = self.model.estimate_raw(images)
        return self.criterion(raw, metadata, parameters=self._plist)

    def __call__(self, images, metadata):
        """Runs one optimisation step; returns (loss (1,), {name: (B,)}) as device tensors."""
        raw = self.model.estimate_raw(images)
        self.opt.zero_grad()
        loss, parts = self.criterion(raw, metadata, parameters=self._plist)
        guard = self._backward_and_reduce(loss)
        self.opt.step(grad_scale=1.0 / self.world, guard_loss=guard)
        self._weights_updated()
        return loss.detach(), parts

    def step_from_store(self, store, pair_ids: torch.Tensor):
        """One step on the pairs `pair_ids` (int64 device tensor) of a device-resident PairStore: the batch is gathered by
        one HIP launch (cd_gather_pairs).  Returns (loss, parts, metadata)."""
        images, metadata = store.batch(pair_ids)
        loss, parts = self(images, metadata)
        return loss, parts, metadata

    # -- the step split at the data-parallel exchange (GraphedFineTuneStep captures the two halves)
    def _grads(self, images, metadata):
        raw = self.model.estimate_raw(images)
        self.opt.zero_grad()
        loss, parts = self.criterion(raw, metadata, parameters=self._plist)
        self._backward(loss)
        guard = loss.detach()
        if self.world > 1:
            self.opt.loss_slot.copy_(guard.reshape(1))
        return guard, parts

    def _update(self, guard):
        if self.world > 1:
            parallel.allreduce_sum_(self.opt.reduce_buffer)
            guard = self.opt.loss_slot
        self.opt.step(grad_scale=1.0 / self.world, guard_loss=guard)
        self._weights_updated()

    @torch.no_grad()
    def evaluate(self, images, metadata):
        """Validation forward: model stays in whatever mode it is in (the reference keeps
        train-mode BatchNorm during validation, depth_fine_tuning.py:241,327-328)."""
        raw = self.model.estimate_raw(images)
        loss, parts = self.criterion(raw, metadata, parameters=self._plist)
        return raw, loss, parts


def _flatten(metadata, prefix=()):
    """(path, tensor) pairs of a nested metadata dict / list, in a fixed order."""
    out = []
    if isinstance(metadata, dict):
        for k in sorted(metadata):
            out += _flatten(metadata[k], prefix + (k,))
    elif isinstance(metadata, (list, tuple)):
        for i, v in enumerate(metadata):
            out += _flatten(v, prefix + (i,))
    elif torch.is_tensor(metadata):
        out.append((prefix, metadata))
    return out


def _clone_tree(metadata):
    if isinstance(metadata, dict):
        return {k: _clone_tree(v) for k, v in metadata.items()}
    if isinstance(metadata, (list, tuple)):
        return type(metadata)(_clone_tree(v) for v in metadata)
    if torch.is_tensor(metadata):
        return metadata.detach().clone().contiguous()
    return metadata


def _store_key(store, pair_ids):
    """What a captured graph's static buffers depend on, for BOTH graph caches: the store's resident buffers (the caching allocator hands a
    freed block's address to the next allocation, and a collected store's id() can be reused: neither alone identifies a store), its
    geometry and size stated explicitly, the record width (PairStore.rebuild_masks changes it) and the batch size."""
    return (id(store), int(store.color.data_ptr()), int(store.flows.data_ptr()), tuple(store.color.shape), len(store),
            int(store.tile_windows.shape[1]), int(pair_ids.numel()))


class GraphedFineTuneStep:
    """A FineTuneStep replayed from a HIP graph.

    One step is ~1300 kernel launches (157 convolutions x {forward, input gradient, weight gradient}, BatchNorm,
    pooling, loss, Adam) issued from Python; at ~20 us of host time each the host, not the MI355X, bounds the step.
    The first `eager_steps` calls per input signature run the plain FineTuneStep (they build the engine plan, time
    the convolution launch shapes, size the workspaces); the next call captures the whole step -- zero-grad, CNN
    forward, fused loss, CNN backward and (single GPU) the guarded Adam update, including the engine's side streams
    -- into one graph with static input buffers, and every later call copies its batch into those buffers and
    replays.  The training trajectory is that of the eager step (capturing executes nothing).  With world > 1 the
    graph ends before the gradient all-reduce by default: the collective and the one Adam launch stay eager.
    CD_AMD_DP_GRAPH_COLLECTIVE=1 captures them too -- RCCL's all-reduce is an ordinary kernel launch on the capturing
    stream (torch's ProcessGroupNCCL supports capture), the step is then ONE replay per rank.  Exercised with a one-rank
    RCCL group on one GPU (tests/test_dp_gpu.py::test_rccl_collective_inside_the_step_graph); opt-in until it has run on
    a multi-GPU node.  A rank that fell back to eager steps (capture failure) while the others replay would desynchronise the
    call sequence: the ranks therefore AGREE after the capture attempt (parallel.all_agree: one eager all-reduce of a flag) and,
    unless every rank holds a graph, all of them drop to the eager exchange together (tests/test_parallel_cpu.py, two gloo ranks,
    one forced to fail).

    If capture fails (a torch / HIP runtime without stream capture) the step stays eager and `self.graphed` is False.
    """

    def __init__(self, step: FineTuneStep, eager_steps: int = 2):
        self.step, self.eager_steps = step, eager_steps
        self._seen, self._graphs, self._store_sig = {}, {}, {}
        self.graphed = None      # None: nothing captured yet; True / False after the first attempt
        self.capture_error = None
        self.graph_collective = step.world > 1 and os.environ.get("CD_AMD_DP_GRAPH_COLLECTIVE", "0") == "1"

    @staticmethod
    def _signature(images, metadata):
        return (tuple(images.shape),) + tuple((path, tuple(t.shape), t.dtype) for path, t in _flatten(metadata))

    def _capture(self, images, metadata):
        st_images, st_meta = images.detach().clone().contiguous(), _clone_tree(metadata)
        dev = images.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            # thread_local: other threads (RCCL's watchdog) may touch the runtime while this thread captures
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                guard, parts = self.step._grads(st_images, st_meta)
                if self.step.world == 1 or self.graph_collective:
                    self.step._update(guard)
        torch.cuda.current_stream(dev).wait_stream(side)
        return {"graph": graph, "images": st_images, "meta": st_meta, "flat": [t for _, t in _flatten(st_meta)],
                "guard": guard, "parts": parts}

    def __call__(self, images, metadata):
        key = self._signature(images, metadata)
        n = self._seen.get(key, 0)
        self._seen[key] = n + 1
        if self.graphed is False or n < self.eager_steps:
            return self.step(images, metadata)
        g = self._graphs.get(key)
        if g is None:
            err = None
            try:
                g = self._capture(images, metadata)
            except Exception as e:   # noqa: BLE001 -- stay on the (equally native) eager path
                err = e
            if self.graph_collective:
                # CAPTURE CONSENSUS (round 6): with the all-reduce inside the graph a rank that replays while another one runs eager
                # steps would desynchronise the collectives.  Capturing executes nothing, so no collective has been issued yet: all
                # ranks agree here (one eager all-reduce(MIN) of a flag, same position in every rank's call sequence -- the first
                # call of this signature beyond the eager ones) whether EVERYBODY has a graph; if not, everybody drops to the eager
                # exchange (graphs without the collective), the path the two-rank tests cover.
                if not parallel.all_agree(err is None, images.device):
                    self.graph_collective = False
                    self.capture_error = (f"{type(err).__name__}: {err}" if err is not None else
                                          "another rank could not capture the step with its all-reduce") + " -> eager exchange on every rank"
                    self._graphs.clear()
                    err, g = None, None
                    try:
                        g = self._capture(images, metadata)        # (now without the collective: it ends before the exchange)
                    except Exception as e:   # noqa: BLE001
                        err = e
            if err is not None:
                self.graphed, self.capture_error = False, f"{type(err).__name__}: {err}"
                torch.cuda.synchronize()
                return self.step(images, metadata)
            self._graphs[key], self.graphed = g, True
        g["images"].copy_(images)
        for dst, (_, src) in zip(g["flat"], _flatten(metadata)):
            dst.copy_(src)
        g["graph"].replay()
        if self.step.world > 1 and not self.graph_collective:
            self.step._update(g["guard"])
        else:
            self.step._weights_updated()     # the update ran inside the replay
        return g["guard"].clone(), {k: v.clone() for k, v in g["parts"].items()}

    def step_from_store(self, store, pair_ids: torch.Tensor):
        """Like FineTuneStep.step_from_store; once the graph of this batch shape exists the pairs are gathered STRAIGHT into
        its static input buffers (no intermediate batch, no copies) and the graph is replayed."""
        # (the record width is part of the key: PairStore.rebuild_masks changes it, and a stale graph's static buffers with it)
        skey = _store_key(store, pair_ids)
        key = self._store_sig.get(skey)
        g = self._graphs.get(key) if key is not None else None
        if g is None:
            images, metadata = store.batch(pair_ids)
            self._store_sig[skey] = self._signature(images, metadata)
            loss, parts = self(images, metadata)
            return loss, parts, metadata
        self._seen[key] = self._seen.get(key, 0) + 1
        store.gather_into(pair_ids, g["images"], g["meta"])
        g["graph"].replay()
        if self.step.world > 1 and not self.graph_collective:
            self.step._update(g["guard"])
        else:
            self.step._weights_updated()
        return g["guard"].clone(), {k: v.clone() for k, v in g["parts"].items()}, g["meta"]

    def evaluate(self, images, metadata):
        return self.step.evaluate(images, metadata)

    def close(self):
        self.step.close()


class GraphedEvaluate:
    """The validation forward of one batch -- CNN forward (train-mode BatchNorm under no_grad, like the reference's sweep,
    depth_fine_tuning.py:241,327-328) + forward-only loss -- replayed from a HIP graph.

    A validation sweep is 179 batches of ~350 launches each (715 pairs, BS4): issued from Python it is bound by the host (~7 ms of
    enqueue per batch for ~6 ms of GPU work), and it is a quarter of an epoch's compute.  After `eager_calls` calls per batch size (they
    build the engine plan and time the launch shapes) the forward + loss of that batch size is captured once; every later batch is
    ONE gather launch straight into the graph's static input buffers (PairStore.gather_into) + ONE replay.  The returned raw output and
    per-pair losses are the graph's static output buffers: consume them (copy, reduce, hand to the writer) before the next call.
    The sweep's short last batch has its own size, is seen once per sweep and stays eager.  Same kernels, same order, same values."""

    def __init__(self, step: FineTuneStep, eager_calls: int = 1):
        self.step, self.eager_calls = step, eager_calls
        self._graphs, self._seen = {}, {}
        self.enabled = os.environ.get("CD_AMD_EVAL_GRAPH", os.environ.get("CD_AMD_STEP_GRAPH", "1")) != "0"
        self.capture_error = None

    def __call__(self, store, pair_ids: torch.Tensor):
        """-> (raw network output (B,2,H,W), {name: (B,)} per-pair losses, metadata of the batch).  The three are the graph's STATIC
        output buffers: consume them (copy, reduce, hand to the writer) before the next call overwrites them."""
        # keyed by the store's resident colour buffer (an address that lives as long as the store's data), not by id(store): a
        # collected store's id can be reused by a new one, whose geometry only coincidentally matches a stale graph
        key = _store_key(store, pair_ids)
        g = self._graphs.get(key)
        if g is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            images, metadata = store.batch(pair_ids)
            if not self.enabled or n < self.eager_calls:
                raw, _, parts = self.step.evaluate(images, metadata)
                return raw, parts, metadata
            try:
                g = self._capture(images, metadata)
            except Exception as e:   # noqa: BLE001 -- the (equally native) eager path
                self.enabled, self.capture_error = False, f"{type(e).__name__}: {e}"
                torch.cuda.synchronize()
                raw, _, parts = self.step.evaluate(images, metadata)
                return raw, parts, metadata
            self._graphs[key] = g
        else:
            store.gather_into(pair_ids, g["images"], g["meta"])
        g["graph"].replay()
        return g["raw"], g["parts"], g["meta"]

    def _capture(self, images, metadata):
        st_images, st_meta = images.detach().clone().contiguous(), _clone_tree(metadata)
        dev = images.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                raw, _, parts = self.step.evaluate(st_images, st_meta)
        torch.cuda.current_stream(dev).wait_stream(side)
        return {"graph": graph, "images": st_images, "meta": st_meta, "raw": raw, "parts": parts}
