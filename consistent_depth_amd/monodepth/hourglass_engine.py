"""Hand-written-HIP execution engine for the Mannequin-Challenge hourglass (forward AND backward).

Executes the same network as consistent_depth_amd/monodepth/hourglass.py (whose nn.Module stays the
parameter / state_dict container) on the gfx950 kernels behind the C ABI:

    conv (all 157)      cd_conv2d_fwd_cfg    MFMA direct convolution; the producer's BatchNorm + ReLU
                                             is applied while loading, the batch statistics of the
                                             raw output are accumulated in the epilogue
    BatchNorm (train)   cd_bn_finalize       NO pass over the activation: the statistics the conv epilogue accumulated become a
                                             per-channel (scale, shift); the buffer keeps the RAW conv output and every consumer
                                             applies relu(raw * scale + shift) while loading (round 2 normalised in place:
                                             0.93 ms of pure HBM traffic per step); running stats updated like PyTorch
    AvgPool2d(2)        cd_avgpool2_fwd
    Upsample x2 + add   cd_upsample2x_add_fwd   (the residual add of every Channels block is fused in)
    backward            cd_bn_relu_bwd, cd_conv2d_wgrad, cd_conv2d_fwd on transposed filters (dgrad),
                        cd_avgpool2_bwd, cd_upsample2x_bwd, cd_add_slice, cd_channel_sum

Inception layout: every inception owns ONE buffer [m1|m2|m3 | b0|o1|o2|o3] (mid activations first, then
the concat output), so its four branch-entry 1x1 convolutions are a single convolution over a contiguous
channel range: X is read once in the forward and dX written once in the backward (PointwiseGroup).

Streams (CD_AMD_ENGINE_STREAMS, default "level"): the full-resolution (skip) side of every Channels block runs on
its own HIP stream, concurrent with the chain through the deeper levels -- that keeps the 256 CUs busy while the
96x56 ... 24x14 levels run kernels with too few workgroups.  ("branch" additionally forks the three k x k branches
of every inception; measured slower.)  All forks/joins are event-based, so a whole training step is capturable
into a HIP graph (consistent_depth_amd/engine.py::GraphedFineTuneStep).

Launch economy: one launch re-packs all filters per step, one writes all weight gradients at the end of the
backward, the BatchNorm passes of adjacent channel slices of an inception share launches ([m1|m2|m3|b0] and
[o1|o2|o3]; their running statistics are views of contiguous buffers), each convolution's launch shape is
timed once per distinct shape (ops/conv.py::tuned_config), and the three k x k branch convolutions of an inception
run as ONE dispatch, forward and input gradient (cd_conv2d_fwd_multi: the branch is a grid dimension; chosen per
inception by timing it against the branches' own launches, ops/conv.py::tuned_multi; bit-identical either way).

No autograd tape is built for the network: the backward pass is the explicit reverse walk of the plan.
Towards PyTorch the engine is ONE autograd node: forward(x) returns pred_d attached to the graph, and
loss.backward() calls `_backward`, which writes every parameter gradient into p.grad (views of the
flat Adam buffer).  The unused `uncertainty_layer` head is not evaluated (its output is discarded by
the reference, mannequin_challenge_model.py:60, and its parameters never receive gradients).

Conv biases in front of a train-mode BatchNorm have an identically zero gradient (the mean subtraction
removes them); the engine leaves those gradients untouched instead of adding round-off noise.
Every parameter gradient is ACCUMULATED into p.grad (torch's convention; the optimiser's zero_grad clears them once
per step), so gradients other autograd nodes put there first -- ParameterLoss with lambda_parameter > 0 -- survive.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from .. import _native
from ..ops import conv as C
from ..ops import layers as L
from . import hourglass as HG

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


class Act:
    """An activation of the plan: channels [coff, coff+C) of `buf`; consumers apply
    relu(v*scale + shift) on load when flagged.  `gbuf` receives d loss / d (activated value)."""

    registry = None  # set while a plan is being built: every Act of the plan is collected here

    def __init__(self, buf, coff, C, relu=False, scale=None, shift=None, needs_grad=True):
        self.buf, self.coff, self.C, self.relu, self.scale, self.shift = buf, coff, C, relu, scale, shift
        self.gbuf = torch.empty_like(buf) if needs_grad else None
        self.gcoff = coff           # channel offset of the gradient inside gbuf (= coff unless gbuf is another activation's buffer)
        self.grad_written = False
        if Act.registry is not None:
            Act.registry.append(self)

    def grad_mode(self) -> bool:
        """accumulate flag for the next gradient contribution (first writer overwrites)."""
        acc = self.grad_written
        self.grad_written = True
        return acc


class ConvUnit:
    def __init__(self, eng, conv_mod, bn_mod, src: Act, dst_buf, dst_coff, stats, mean_invstd, bn_scale=None, bn_shift=None):
        """bn_scale / bn_shift: the (ctot,) apply-on-load arrays of the destination buffer (cd_bn_finalize fills this unit's slice)."""
        self.eng, self.conv, self.bn, self.src = eng, conv_mod, bn_mod, src
        self.ks, self.cin, self.cout = conv_mod.kernel_size[0], conv_mod.in_channels, conv_mod.out_channels
        self.dst_buf, self.dst_coff, self.stats, self.mi = dst_buf, dst_coff, stats, mean_invstd
        self.bn_scale, self.bn_shift = bn_scale, bn_shift
        if bn_mod is not None and bn_scale is not None:      # apply-on-load: the buffer keeps the raw convolution output
            sc, sh = bn_scale[dst_coff:dst_coff + self.cout], bn_shift[dst_coff:dst_coff + self.cout]
        elif bn_mod is not None and bn_mod.affine:            # normalise-in-place mode (CD_AMD_BN_APPLY=0): x_hat in the buffer, the stem's affine on load
            sc, sh = bn_mod.weight, bn_mod.bias
        else:
            sc = sh = None
        self.out = Act(dst_buf, dst_coff, self.cout, relu=bn_mod is not None, scale=sc, shift=sh, needs_grad=False)
        self.wgrad_ws = self.sums = None  # views into the plan's arenas (HourglassEngine._carve_arenas)
        self.wgrad_batched = False        # True: this unit's weight gradient is part of the plan's WgradTable (one launch per kernel
                                          # class at the end of the backward pass) -- backward() does not launch it
        self.bn_fused = False             # True: the owning inception runs the BN passes of its three branch outputs jointly
        self.dgrad_merged = False         # True: the owning inception computes this unit's input gradient in ONE dispatch with its siblings
        self.pk, self.pkT = eng.packed(conv_mod)
        # launch shapes, timed once per distinct convolution shape (ops/conv.py::tuned_config)
        N, _, H, W = dst_buf.shape
        self.cfg_f = C.tuned_config(self.ks, self.cin, self.cout, N, H, W, dst_buf.device, affine_in=src.scale is not None,
                                    relu_in=src.relu, stats=bn_mod is not None, x_ctot=src.buf.shape[1], y_ctot=dst_buf.shape[1])
        self.cfg_d = C.tuned_config(self.ks, self.cout, self.cin, N, H, W, dst_buf.device, accumulate=True,
                                    x_ctot=dst_buf.shape[1], y_ctot=src.buf.shape[1])

    def fwd_member(self, training):
        """This unit's forward convolution as a member of a merged dispatch (ops/conv.py::conv2d_multi)."""
        s = self.src
        return dict(x=s.buf, packed_w=self.pk, Cin=self.cin, Cout=self.cout, ks=self.ks, bias=self.conv.bias, x_coff=s.coff, out=self.dst_buf,
                    y_coff=self.dst_coff, in_scale=s.scale, in_shift=s.shift, in_relu=s.relu,
                    stats=self.stats.view(-1) if (self.bn is not None and training) else None)

    def dgrad_member(self, gbuf, g_coff, accumulate):
        s = self.src
        return dict(x=gbuf, packed_w=self.pkT, Cin=self.cout, Cout=self.cin, ks=self.ks, x_coff=g_coff, out=s.gbuf, y_coff=s.gcoff,
                    accumulate=accumulate)

    def forward(self, training):
        s = self.src
        C.conv2d(s.buf, self.pk, self.cin, self.cout, self.ks, bias=self.conv.bias, x_coff=s.coff, out=self.dst_buf,
                 y_coff=self.dst_coff, in_scale=s.scale, in_shift=s.shift, in_relu=s.relu,
                 stats=self.stats.view(-1) if (self.bn is not None and training) else None, cfg=self.cfg_f)
        if self.bn is not None and not self.bn_fused:
            affine = self.bn.affine
            self.eng.bn_forward(self.dst_buf, self.dst_coff, self.cout, self.stats, self.mi, (self.bn.running_mean, self.bn.running_var),
                                training, self.bn_scale, self.bn_shift, gamma=self.bn.weight if affine else None,
                                beta=self.bn.bias if affine else None)

    def backward(self, gbuf, g_coff):
        """gbuf[:, g_coff:+cout] holds d loss / d (activated output); on return the parameter grads are
        written and the source activation's gradient received this unit's contribution."""
        s = self.src
        if self.bn_fused:
            pass   # the inception already turned gbuf into the gradient w.r.t. the raw convolution output
        elif self.bn is not None:
            affine = self.bn.affine
            L.bn_relu_bwd(gbuf, g_coff, self.dst_buf, self.dst_coff, self.cout, self.mi, self.sums,
                          gamma=self.bn.weight if affine else None, beta=self.bn.bias if affine else None,
                          dgamma=_grad_of(self.bn.weight) if affine else None,
                          dbeta=_grad_of(self.bn.bias) if affine else None, sums_prezeroed=True, scale=self.bn_scale, shift=self.bn_shift)
        else:
            L.channel_sum(gbuf, g_coff, self.cout, _grad_of(self.conv.bias), accumulate=True)
        # the partial sums stay packed in the arena; plan["unpack"] writes every weight gradient at the end of the backward
        if not self.wgrad_batched:
            self.eng.on_wgrad_stream(lambda: C.conv2d_wgrad(
                s.buf, gbuf, self.cin, self.cout, self.ks, None, self.wgrad_ws, x_coff=s.coff, dy_coff=g_coff, in_scale=s.scale,
                in_shift=s.shift, in_relu=s.relu, prezeroed=True))
        if s.gbuf is not None and not self.dgrad_merged:
            C.conv2d(gbuf, self.pkT, self.cout, self.cin, self.ks, x_coff=g_coff, out=s.gbuf, y_coff=s.gcoff,
                     accumulate=s.grad_mode(), cfg=self.cfg_d)


class _Member:
    """One of the fused 1x1 convolutions (keeps its own parameters, BN module and weight-gradient workspace)."""

    def __init__(self, conv_mod, bn_mod, coff):
        self.conv, self.bn, self.coff = conv_mod, bn_mod, coff
        self.ks, self.cin, self.cout = 1, conv_mod.in_channels, conv_mod.out_channels
        self.wgrad_ws = self.sums = None


class PointwiseGroup:
    """The four branch-entry 1x1 convolutions of an inception as ONE convolution X -> P[:, 0:ctot]."""

    def __init__(self, eng, members, src: Act, P, Pg, stats, mean_invstd, filt, filtT, running, bn_scale, bn_shift):
        self.eng, self.members, self.src, self.P, self.Pg, self.stats, self.mi = eng, members, src, P, Pg, stats, mean_invstd
        self.bn_scale, self.bn_shift = bn_scale, bn_shift
        self.running = running            # (running_mean, running_var) of the members, contiguous in member order
        self.ctot = sum(m.cout for m in members)
        self.cin = members[0].cin
        self.cout, self.ks = self.ctot, 1          # the group is ONE unit for the arenas (wgrad workspace, BN sums)
        self.wgrad_ws = self.sums = None
        self._filt, self._filtT = filt, filtT
        self._bias = torch.empty(self.ctot, device=P.device)
        self._bias_versions = None
        N, _, H, W = P.shape
        self.cfg_f = C.tuned_config(1, self.cin, self.ctot, N, H, W, P.device, affine_in=src.scale is not None, relu_in=src.relu,
                                    stats=True, x_ctot=src.buf.shape[1], y_ctot=P.shape[1])
        self.cfg_d = C.tuned_config(1, self.ctot, self.cin, N, H, W, P.device, accumulate=True, x_ctot=Pg.shape[1],
                                    y_ctot=src.buf.shape[1])

    def _fused_bias(self):
        # Refreshed on EVERY forward: FlatAdam updates the member biases through raw pointers, which tensor version counters never
        # see, so a cached copy would go stale -- harmless under a train-mode BatchNorm (which cancels the bias) but wrong in eval
        # mode and as soon as the biases get a gradient (lambda_parameter > 0).  Round 6: the copies of ALL groups of a plan are one
        # cd_copy_segments launch at the top of the forward (_BiasTable; 22 torch.cat launches per forward before).
        return self._bias

    def forward(self, training):
        s, pk = self.src, self.eng._pack.view(self._filt)
        C.conv2d(s.buf, pk, self.cin, self.ctot, 1, bias=self._fused_bias(), x_coff=s.coff, out=self.P, y_coff=0,
                 in_scale=s.scale, in_shift=s.shift, in_relu=s.relu, stats=self.stats.view(-1) if training else None,
                 cfg=self.cfg_f)
        self.eng.bn_forward(self.P, 0, self.ctot, self.stats, self.mi, self.running, training, self.bn_scale, self.bn_shift)   # [m1|m2|m3|b0]

    def backward(self):
        s = self.src
        L.bn_relu_bwd(self.Pg, 0, self.P, 0, self.ctot, self.mi, self.sums, sums_prezeroed=True, scale=self.bn_scale, shift=self.bn_shift)
        # ONE weight-gradient GEMM for the four filters (X is read once), then split the rows
        self.eng.on_wgrad_stream(lambda: C.conv2d_wgrad(
            s.buf, self.Pg, self.cin, self.ctot, 1, None, self.wgrad_ws, x_coff=s.coff, dy_coff=0, in_scale=s.scale,
            in_shift=s.shift, in_relu=s.relu, prezeroed=True))
        if s.gbuf is not None:
            C.conv2d(self.Pg, self.eng._pack.view(self._filtT), self.ctot, self.cin, 1, x_coff=0, out=s.gbuf, y_coff=s.gcoff,
                     accumulate=s.grad_mode(), cfg=self.cfg_d)


class _BiasTable:
    """member bias -> fused bias vector of every PointwiseGroup of a plan, as ONE launch (cd_copy_segments)."""

    def __init__(self, groups, device):
        self.groups, self.device, self._key, self._table = groups, device, None, None

    def run(self):
        if not self.groups:
            return
        key = tuple(m.conv.bias.data_ptr() for g in self.groups for m in g.members)
        if key != self._key:        # (first use, or the parameters were re-homed -- FlatAdam moves them into its flat buffer)
            rows = []
            for g in self.groups:
                off = 0
                for m in g.members:
                    rows.append((m.conv.bias.data_ptr(), g._bias.data_ptr() + 4 * off, m.cout))
                    off += m.cout
                assert off == g.ctot
            self._table, self._key = torch.tensor(rows, dtype=torch.int64, device=self.device), key
        _native.check(_native.lib().cd_copy_segments(self._table.data_ptr(), self._table.shape[0], _native.stream_ptr(self.device)),
                      "cd_copy_segments")


def _groups_of(steps):
    out = []
    for st in steps:
        if st.kind == "inception":
            out.append(st.group)
        elif st.kind == "channels":
            out += _groups_of(st.flat) + _groups_of(st.up)
    return out


def _grad_of(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class _Node:
    """Non-conv plan steps."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)


class HourglassEngine:
    def __init__(self, net: HG.HourglassModel):
        self.net = net
        self.device = next(net.parameters()).device
        self._plans = {}
        # every filter (forward + flipped/transposed dgrad twin) is re-packed by ONE launch per forward
        self._pack = C.PackTable(self.device)
        self._pack_index = {}
        self._group_filters = {}
        self._bn_flat = {}
        grouped = set()
        for inc in net.modules():
            if isinstance(inc, HG.Inception):
                entry = [br[0] for br in list(inc.convs)[1:]] + [inc.convs[0][0]]   # [m1, m2, m3, b0]
                ctot, cin = sum(c.out_channels for c in entry), entry[0].in_channels
                f, fT = self._pack.new_filter(ctot, cin, 1), self._pack.new_filter(cin, ctot, 1)
                off = 0
                for c in entry:
                    self._pack.source(f, c.weight, False, oc_off=off)    # forward: concatenated output channels
                    self._pack.source(fT, c.weight, True, ic_off=off)    # dgrad: concatenated input channels
                    off += c.out_channels
                    grouped.add(id(c))
                self._group_filters[id(inc)] = (f, fT)
                # the BatchNorms of adjacent channel slices share launches: their running statistics become views of
                # contiguous buffers (entry convs [m1|m2|m3|b0], branch outputs [o1|o2|o3]); names / state_dict unchanged
                self._bn_flat[id(inc)] = (self._rehome_running_stats([br[1] for br in list(inc.convs)[1:]] + [inc.convs[0][1]]),
                                          self._rehome_running_stats([br[4] for br in list(inc.convs)[1:]]))
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d) and m is not net.uncertainty_layer[0] and id(m) not in grouped:
                self._pack_index[id(m)] = (self._pack.add(m.weight, False), self._pack.add(m.weight, True))
        self._pack.build()
        # streams: one per Channels level for its full-resolution side, and three branch streams per parent stream
        mode = os.environ.get("CD_AMD_ENGINE_STREAMS", "level")   # none | branch | level | both (eager, round start: 73 | 80 | 86 | 76 pairs/s;
        # with graph replay: none 100.6, level 109.6)
        self.use_branch_streams = mode in ("branch", "both")
        self.use_level_streams = mode in ("level", "both")
        self._level_streams = {lvl: torch.cuda.Stream(device=self.device) for lvl in (1, 2, 3, 4)}
        self._branch_streams = {}
        # Weight gradients are off the critical path (only the optimiser needs them), so they could trail the
        # input-gradient chain on their own stream.  Measured on MI355X: SLOWER (graph replay 109.4 -> 102.9 pairs/s, eager
        # 107.5 -> 105.9): the wgrad kernels fill every CU on their own and only take cycles from the dgrad chain, and the
        # forked graph doubles the host cost of a replay.  Opt-in for experiments: CD_AMD_ENGINE_WGRAD_STREAM=1.
        self._wgrad_stream = torch.cuda.Stream(device=self.device) \
            if mode != "none" and os.environ.get("CD_AMD_ENGINE_WGRAD_STREAM", "0") == "1" else None
        self._wgrad_pending = False
        # BatchNorm as (scale, shift) applied by the consumers while loading (default), or round 2's in-place normalisation pass
        # (CD_AMD_BN_APPLY=0: kept for A/B measurements on one box)
        self.bn_apply = os.environ.get("CD_AMD_BN_APPLY", "1") != "0"
        # CD_AMD_WGRAD_BATCH=1: the k x k weight gradients of a step as one launch per kernel class at the end of the backward pass
        # (ops/conv.py::WgradTable; bit-identical results).  Measured in round 4 and left OFF: 67 launches become 5, but the kernel
        # time of a step barely moves (5.21 -> 5.07 ms: the deep levels' gradients are bound by the latency of their few workgroups'
        # own tile loops, not by idle CUs that other gradients could fill) and the step loses the overlap of the side streams'
        # gradients with the main chain (170.1 vs 171.5 pairs/s, two runs each; profiles/wgrad_batch_r04.txt).
        self.wgrad_batch = os.environ.get("CD_AMD_WGRAD_BATCH", "0") == "1"
        # nn.BatchNorm2d counts its train-mode forwards (training steps AND the reference's train-mode validation batches);
        # momentum is fixed so nothing reads the counters, but they are part of the checkpoint the reference writes
        self._batch_counters = [m.num_batches_tracked for m in net.modules()
                                if isinstance(m, torch.nn.BatchNorm2d) and m.num_batches_tracked is not None]

    def _bias_table(self, plan):
        if "bias_table" not in plan:
            plan["bias_table"] = _BiasTable(_groups_of(plan["steps"]), self.device)
        return plan["bias_table"]

    def _advance_batch_counters(self):
        """num_batches_tracked += 1 of every BatchNorm2d: one launch over a device table of the counters' addresses (torch._foreach_add_
        was two multi-tensor launches inside the captured step).  The table follows the counters if a state-dict load re-binds them."""
        ptrs = tuple(t.data_ptr() for t in self._batch_counters)
        if ptrs != getattr(self, "_counter_key", None):
            if not all(t.is_cuda and t.dtype == torch.int64 for t in self._batch_counters):
                raise RuntimeError("BatchNorm batch counters must be int64 tensors on the HIP device")
            self._counter_table, self._counter_key = torch.tensor(ptrs, dtype=torch.int64, device=self.device), ptrs
        _native.check(_native.lib().cd_counters_add(self._counter_table.data_ptr(), len(ptrs), 1, _native.stream_ptr(self.device)),
                      "cd_counters_add")

    def _rehome_running_stats(self, bns):
        n = sum(b.num_features for b in bns)
        rm, rv = torch.empty(n, device=self.device), torch.empty(n, device=self.device)
        off = 0
        with torch.no_grad():
            for b in bns:
                assert not b.affine and b.track_running_stats
                c = b.num_features
                rm[off:off + c].copy_(b.running_mean)
                rv[off:off + c].copy_(b.running_var)
                b.running_mean, b.running_var = rm[off:off + c], rv[off:off + c]
                off += c
        return rm, rv

    def packed(self, conv_mod):
        i, j = self._pack_index[id(conv_mod)]
        return self._pack.view(i), self._pack.view(j)

    # ------------------------------------------------------------------ stream helpers
    def _branches_of(self, parent):
        key = parent.cuda_stream
        if key not in self._branch_streams:
            self._branch_streams[key] = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        return self._branch_streams[key]

    def _fork_join(self, jobs):
        """Run the callables concurrently on the branch streams of the current stream (fork/join with events)."""
        if not self.use_branch_streams or len(jobs) < 2:
            for j in jobs:
                j()
            return
        cur = torch.cuda.current_stream(self.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        for j, st in zip(jobs, self._branches_of(cur)):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                j()
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)

    def bn_forward(self, buf, coff, C, stats, mi, running, training, scale, shift, gamma=None, beta=None):
        """Train-mode BatchNorm (or eval: the running statistics) of C adjacent channels WITHOUT touching the activation: ONE tiny
        launch turns the statistics into the (scale, shift) the consumers apply while loading the raw convolution output."""
        rm, rv = running
        cnt = float(buf.shape[0] * buf.shape[2] * buf.shape[3])
        if not training:   # synthesised sums in slot 0 (the others stay zero); nothing is updated
            stats[0, coff:coff + C, 0] = rm.double() * cnt
            stats[0, coff:coff + C, 1] = (rv.double() + rm.double() * rm.double()) * cnt
        if scale is None:  # CD_AMD_BN_APPLY=0 (A/B): round 2's in-place normalisation, x_hat in the buffer
            L.bn_normalize(buf, coff, C, stats, mi, BN_EPS, rm if training else None, rv if training else None, BN_MOMENTUM)
        elif training:
            L.bn_finalize(stats, coff, C, cnt, mi, scale, shift, BN_EPS, gamma, beta, rm, rv, BN_MOMENTUM)
        else:
            L.bn_finalize(stats, coff, C, cnt, mi, scale, shift, BN_EPS, gamma, beta)

    def on_wgrad_stream(self, job):
        """Run `job` (a weight-gradient launch sequence) on the wgrad stream, ordered after everything enqueued so far
        on the current stream; `_backward` joins before returning."""
        ws = self._wgrad_stream
        if ws is None:
            job()
            return
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        ws.wait_event(ready)
        with torch.cuda.stream(ws):
            job()
        self._wgrad_pending = True

    def _on_side(self, level, job):
        """Run `job` on the level's side stream, forked from the current stream; returns the completion event
        (None when streams are off: the job has simply run)."""
        if not self.use_level_streams:
            job()
            return None
        cur, st = torch.cuda.current_stream(self.device), self._level_streams[level]
        fork = torch.cuda.Event()
        fork.record(cur)
        st.wait_event(fork)
        with torch.cuda.stream(st):
            job()
        done = torch.cuda.Event()
        done.record(st)
        return done

    def _join(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    # ------------------------------------------------------------------ plan construction (a tree of steps)
    def _new(self, N, Ch, H, W):
        return torch.empty(N, Ch, H, W, dtype=torch.float32, device=self.device)

    def _stats(self, plan, channels):
        """(STAT_SLOTS, channels, 2) fp64 view of the plan's statistics arena (zeroed by ONE memset per forward)."""
        a, n = plan["stats_used"], L.STAT_SLOTS * channels
        if a + n > plan["stats_arena"].shape[0]:
            raise RuntimeError("statistics arena too small")
        plan["stats_used"] = a + n
        return plan["stats_arena"][a:a + n].view(L.STAT_SLOTS, channels, 2)

    def _inception(self, plan, steps, mod: HG.Inception, x: Act, N, H, W) -> Act:
        c_in, cfg = HG.INCEPTION[mod.kind]
        a0 = cfg[0][0]
        outs = [a0] + [c[2] for c in cfg[1:]]
        mids = [c[1] for c in cfg[1:]]
        M, Co = sum(mids), sum(outs)
        P = self._new(N, M + Co, H, W)            # [m1|m2|m3 | b0|o1|o2|o3]
        Pg = torch.empty_like(P)
        stats = self._stats(plan, M + Co)
        mi = torch.zeros(M + Co, 2, device=self.device)
        sc = sh = None
        if self.bn_apply:
            sc, sh = torch.ones(M + Co, device=self.device), torch.zeros(M + Co, device=self.device)   # apply-on-load BatchNorm of P
        sl = (lambda t, a, n: t[a:a + n]) if self.bn_apply else (lambda t, a, n: None)
        members, moff = [], 0
        for i, br in enumerate(list(mod.convs)[1:]):
            members.append(_Member(br[0], br[1], moff))
            moff += mids[i]
        members.append(_Member(mod.convs[0][0], mod.convs[0][1], M))
        filt, filtT = self._group_filters[id(mod)]
        run_entry, run_out = self._bn_flat[id(mod)]
        group = PointwiseGroup(self, members, x, P, Pg, stats, mi, filt, filtT, run_entry, sc, sh)
        units, ooff, moff = [], M + a0, 0
        for i, br in enumerate(list(mod.convs)[1:]):
            mid = Act(P, moff, mids[i], relu=True, scale=sl(sc, moff, mids[i]), shift=sl(sh, moff, mids[i]), needs_grad=False)
            mid.gbuf = Pg
            units.append((ConvUnit(self, br[3], br[4], mid, P, ooff, stats, mi, sc, sh), Pg, ooff))
            units[-1][0].bn_fused = True
            ooff += outs[i + 1]
            moff += mids[i]
        out = Act(P, M, Co, relu=True, scale=sl(sc, M, Co), shift=sl(sh, M, Co), needs_grad=False)
        out.gbuf = Pg
        # ONE dispatch for the three k x k branches (largest filter first), forward and input gradient, where it is timed faster than
        # their own launches (ops/conv.py::tuned_multi; same bits either way): a single branch at 96x56 and below launches fewer
        # workgroups than the chip has CUs, and kernels of different streams do not share it (profiles/branch_overlap_r04.txt)
        order = sorted(range(len(units)), key=lambda i: -units[i][0].ks)
        us = [units[i] for i in order]
        multi_f = C.tuned_multi([u.fwd_member(True) for u, _, _ in us], [u.cfg_f for u, _, _ in us])
        multi_d = C.tuned_multi([u.dgrad_member(g, o, False) for u, g, o in us], [u.cfg_d for u, _, _ in us])
        if multi_d is not None:
            for u, _, _ in units:
                u.dgrad_merged = True
        steps.append(_Node("inception", group=group, units=units, out=out, src=x, P=P, Pg=Pg, stats=stats, mi=mi, bn_scale=sc, bn_shift=sh,
                           bn_coff=M + a0, bn_C=sum(outs[1:]), bn_running=run_out, bn_sums=None, merged=us, multi_f=multi_f, multi_d=multi_d))
        plan["convs"] += [group] + [u for u, _, _ in units]
        return out

    def _sequence(self, plan, steps, seq, x: Act, N, H, W):
        """Appends the steps of a Sequential of pool / inception / channels [/ up] to `steps`.
        Returns (Act, H, W, pending_up) where pending_up says the sequence ended with an upsample that the
        caller fuses with its residual add."""
        pending_up = False
        for m in seq:
            if isinstance(m, torch.nn.AvgPool2d):
                y = Act(self._new(N, x.C, H // 2, W // 2), 0, x.C)
                steps.append(_Node("pool", src=x, out=y))
                x, H, W = y, H // 2, W // 2
            elif isinstance(m, HG.Inception):
                x = self._inception(plan, steps, m, x, N, H, W)
            elif isinstance(m, HG.Channels):
                x = self._channels(plan, steps, m, x, N, H, W)
            elif isinstance(m, torch.nn.UpsamplingBilinear2d):
                pending_up = True
            else:
                raise TypeError(f"unexpected module {type(m)}")
        return x, H, W, pending_up

    def _channels(self, plan, steps, mod: HG.Channels, x: Act, N, H, W) -> Act:
        sides = list(mod.list)
        up_side = 0 if isinstance(sides[0][-1], torch.nn.UpsamplingBilinear2d) else 1
        # Both sides read x and both add into its gradient.  The backward passes of the two sides run concurrently (the full-resolution
        # side on the level's stream); x's gradient gets the full-resolution side's contribution first (its last kernel, the first
        # writer: it overwrites) and the pooled side's LAST kernel -- the adjoint of the AvgPool2d the side starts with -- waits for
        # the other stream and accumulates (_run_backward).  Rounds 1-5 gave the full-resolution side an alias of x with its own gradient
        # buffer and added the two at the join: a 528 MB pass at 384x224 (0.2 ms of a 22 ms step) for the sake of overlapping one
        # 60 us kernel.  Same sum, same order as the single-stream C engine (csrc/hourglass.hip): bit-identical.
        flat_steps, up_steps = [], []
        flat, _, _, _ = self._sequence(plan, flat_steps, sides[1 - up_side], x, N, H, W)
        lo, h, w, pending = self._sequence(plan, up_steps, sides[up_side], x, N, H, W)
        assert pending and up_steps[0].kind == "pool" and up_steps[0].src is x
        # out = up(lo) + hi.  Its gradient IS hi's gradient: out's consumers write it straight into hi's slice of the concat gradient
        # buffer (rounds 1-5 gave `out` its own gradient buffer and copied it: a 176 MB pass at 384x224); the bilinear adjoint reads
        # it there before the full-resolution side's BatchNorm backward overwrites it in place (same stream order as before).
        out = Act(self._new(N, flat.C, H, W), 0, flat.C, needs_grad=False)
        out.gbuf, out.gcoff = flat.gbuf, flat.gcoff
        steps.append(_Node("channels", level=mod.level, flat=flat_steps, up=up_steps, lo=lo, hi=flat, out=out, x=x))
        return out

    def _build(self, N, H, W):
        if H % HG.ALIGN or W % HG.ALIGN:
            raise ValueError(f"hourglass input must be a multiple of {HG.ALIGN} in both dimensions, got {H}x{W}")
        net = self.net
        Act.registry = []
        plan = {"steps": [], "convs": [], "stats_used": 0,
                "stats_arena": torch.zeros(16384 * L.STAT_SLOTS, 2, dtype=torch.float64, device=self.device)}
        plan["x"] = self._new(N, 3, H, W)
        x_in = Act(plan["x"], 0, 3, needs_grad=False)
        stem_buf = self._new(N, 128, H, W)
        stem = ConvUnit(self, net.seq[0], net.seq[1], x_in, stem_buf, 0, self._stats(plan, 128), torch.zeros(128, 2, device=self.device),
                        torch.ones(128, device=self.device) if self.bn_apply else None,
                        torch.zeros(128, device=self.device) if self.bn_apply else None)
        stem.out.gbuf = torch.empty_like(stem_buf)
        plan["steps"].append(_Node("conv", unit=stem, gbuf=stem.out.gbuf, g_coff=0))
        plan["convs"].append(stem)
        feat = self._channels(plan, plan["steps"], net.seq[3], stem.out, N, H, W)
        plan["pred"] = self._new(N, 1, H, W)
        head = ConvUnit(self, net.pred_layer, None, feat, plan["pred"], 0, None, None)
        plan["dpred"] = torch.empty_like(plan["pred"])
        plan["steps"].append(_Node("conv", unit=head, gbuf=plan["dpred"], g_coff=0))
        plan["convs"].append(head)
        plan["acts"], Act.registry = Act.registry, None
        self._carve_arenas(plan)
        return plan

    def _carve_arenas(self, plan):
        """One arena for all wgrad workspaces (per-workgroup partial sums, written whole by every launch: never zeroed) and
        one for all BN-backward sums (one memset per backward)."""
        units = plan["convs"]
        sizes = [(C.wgrad_workspace_floats(u.cout, u.cin, u.ks) + 63) // 64 * 64 for u in units]
        plan["wgrad_arena"] = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
        plan["sums_arena"] = torch.zeros(sum(2 * u.cout for u in units), dtype=torch.float64, device=self.device)
        o = so = 0
        for step in self._all_steps(plan["steps"]):
            if step.kind == "inception":   # the three branch units' sums are consecutive in the arena: one joint view
                us = [u for u, _, _ in step.units]
                i0 = units.index(us[0])
                assert units[i0:i0 + 3] == us
                s0 = sum(2 * u.cout for u in units[:i0])
                step.bn_sums = plan["sums_arena"][s0:s0 + 2 * step.bn_C]
        unpack = plan["unpack"] = C.UnpackTable(self.device)
        plan["wgrad_table"] = C.WgradTable(self.device)
        # (unit -> the buffer / channel offset its backward receives the output gradient in)
        gbuf_of = {}
        for step in self._all_steps(plan["steps"]):
            if step.kind == "conv":
                gbuf_of[step.unit] = (step.gbuf, step.g_coff)
            elif step.kind == "inception":
                for u, gb, gc in step.units:
                    gbuf_of[u] = (gb, gc)
        N, _, H, W = plan["x"].shape
        for u, n in zip(units, sizes):
            u.wgrad_ws = plan["wgrad_arena"][o:o + n]
            u.sums = plan["sums_arena"][so:so + 2 * u.cout]
            o += n
            so += 2 * u.cout
            h, w = (u.P.shape[2:] if isinstance(u, PointwiseGroup) else u.dst_buf.shape[2:])
            layout = C.wgrad_plan(u.cout, u.cin, u.ks, N, h, w)
            if self.wgrad_batch and isinstance(u, ConvUnit) and u.ks >= 3 and u in gbuf_of:
                # Deferred AND batched: the operands -- the producer's raw output with its (scale, shift), and the gradient w.r.t. this
                # unit's raw output, final once the unit's backward has run -- are buffers of the plan that nothing overwrites before
                # the end of the backward pass, so the gradient can be computed there, together with all the others of its class.
                gb, gc = gbuf_of[u]
                s = u.src
                u.wgrad_batched = plan["wgrad_table"].add(s.buf, gb, u.cin, u.cout, u.ks, u.wgrad_ws, x_coff=s.coff, dy_coff=gc,
                                                          in_scale=s.scale, in_shift=s.shift, in_relu=s.relu)
            if isinstance(u, PointwiseGroup):   # the fused gradient's rows belong to the members' weights
                for m in u.members:
                    unpack.add(u.wgrad_ws, (lambda m=m: _grad_of(m.conv.weight)), u.cin, 1, layout, row0=m.coff, cout_total=u.cout)
            else:
                unpack.add(u.wgrad_ws, (lambda u=u: _grad_of(u.conv.weight)), u.cin, u.ks, layout)

    def _all_steps(self, steps):
        for st in steps:
            yield st
            if st.kind == "channels":
                yield from self._all_steps(st.flat)
                yield from self._all_steps(st.up)

    def plan(self, N, H, W):
        # the arithmetic mode of the convolutions (cd_set_conv_arith) selects different kernels with different packed
        # weight-gradient layouts and tuned launch shapes: a plan belongs to the mode it was built under
        key = (N, H, W, _native.lib().cd_get_conv_arith())
        if key not in self._plans:
            self._plans[key] = self._build(N, H, W)
        return self._plans[key]

    # ------------------------------------------------------------------ execution
    def _run_forward(self, steps, training):
        for step in steps:
            if step.kind == "conv":
                step.unit.forward(training)
            elif step.kind == "inception":
                step.group.forward(training)
                if step.multi_f is not None:
                    if not C.conv2d_multi([u.fwd_member(training) for u, _, _ in step.merged], step.multi_f):
                        raise RuntimeError("merged branch dispatch refused after it was timed")
                else:
                    self._fork_join([(lambda u=u: u.forward(training)) for u, _, _ in step.units])
                self.bn_forward(step.P, step.bn_coff, step.bn_C, step.stats, step.mi, step.bn_running, training, step.bn_scale, step.bn_shift)   # [o1|o2|o3]
            elif step.kind == "pool":
                s = step.src
                L.avgpool2_fwd(s.buf, s.coff, s.C, step.out.buf, 0, in_scale=s.scale, in_shift=s.shift, in_relu=s.relu)
            elif step.kind == "channels":
                done = self._on_side(step.level, lambda: self._run_forward(step.flat, training))
                self._run_forward(step.up, training)
                self._join(done)
                lo, hi = step.lo, step.hi
                L.upsample2x_add_fwd(lo.buf, lo.coff, lo.C, step.out.buf, 0, hi=hi.buf, hi_coff=hi.coff, lo_relu=lo.relu,
                                     hi_relu=hi.relu, lo_scale=lo.scale, lo_shift=lo.shift, hi_scale=hi.scale,
                                     hi_shift=hi.shift)

    def _run_backward(self, steps, before_last=None):
        for i, step in enumerate(reversed(steps)):
            if before_last is not None and i == len(steps) - 1:
                before_last()
            if step.kind == "conv":
                step.unit.backward(step.gbuf, step.g_coff)
            elif step.kind == "inception":
                # the concat output's gradient is complete: k x k convolutions first (they fill the gradient of the
                # mid activations), then the fused entry convolution
                L.bn_relu_bwd(step.Pg, step.bn_coff, step.P, step.bn_coff, step.bn_C, step.mi, step.bn_sums, sums_prezeroed=True,
                              scale=step.bn_scale, shift=step.bn_shift)
                self._fork_join([(lambda u=u, g=gbuf, o=g_coff: u.backward(g, o)) for u, gbuf, g_coff in step.units])
                if step.multi_d is not None:     # the three input gradients (each the only writer of its mid activation's gradient)
                    if not C.conv2d_multi([u.dgrad_member(g, o, u.src.grad_mode()) for u, g, o in step.merged], step.multi_d):
                        raise RuntimeError("merged branch dispatch refused after it was timed")
                step.group.backward()
            elif step.kind == "pool":
                s = step.src
                L.avgpool2_bwd(step.out.gbuf, step.out.gcoff, s.gbuf, s.gcoff, s.C, accumulate=s.grad_mode())
            elif step.kind == "channels":
                lo, hi, o = step.lo, step.hi, step.out
                # d(up(lo) + hi) / d hi is the identity: the sum's gradient was written straight into hi's gradient slice (_channels)
                assert o.gbuf is hi.gbuf and o.gcoff == hi.gcoff
                hi.grad_written = True
                L.upsample2x_bwd(o.gbuf, o.gcoff, lo.gbuf, lo.gcoff, lo.C, accumulate=lo.grad_mode())
                # the full-resolution side is the first writer of x's gradient; the pooled side's last kernel (the AvgPool2d adjoint into
                # x) is enqueued behind the join and accumulates
                done = self._on_side(step.level, lambda: self._run_backward(step.flat))
                self._run_backward(step.up, before_last=lambda: self._join(done))

    @torch.no_grad()
    def _forward(self, x: torch.Tensor, need_grad: bool) -> torch.Tensor:
        N, _, H, W = x.shape
        plan = self.plan(N, H, W)
        plan["x"].copy_(x)
        _native.zero_(plan["stats_arena"])
        self._pack.run()
        self._bias_table(plan).run()
        self._run_forward(plan["steps"], self.net.training)
        if self.net.training and self._batch_counters:
            self._advance_batch_counters()
        self._last = plan
        return plan["pred"]

    @torch.no_grad()
    def _backward(self, dpred: torch.Tensor):
        plan = self._last
        plan["dpred"].copy_(dpred.reshape(plan["dpred"].shape))
        _native.zero_(plan["sums_arena"])
        for a in plan["acts"]:  # first gradient contribution overwrites, later ones accumulate
            a.grad_written = False
        self._run_backward(plan["steps"])
        if self._wgrad_pending:   # join: the parameter gradients are complete when this returns (stream order)
            torch.cuda.current_stream(self.device).wait_stream(self._wgrad_stream)
            self._wgrad_pending = False
        self._finish_wgrads(plan)

    @staticmethod
    def _finish_wgrads(plan):
        plan["wgrad_table"].run()     # the k x k weight gradients of the whole network: one launch per kernel class
        plan["unpack"].run()          # every weight gradient (157 tensors) written by one launch

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (N,3,H,W) -> pred_d (N,1,H,W) (log depth), attached to autograd when grad is enabled."""
        x = x.to(self.device, torch.float32).contiguous()
        if torch.is_grad_enabled():
            anchor = next(self.net.parameters())
            return _EngineFn.apply(anchor, self, x)
        return self._forward(x, need_grad=False).clone()


class BlockRunner:
    """ONE inception of the network as a stand-alone plan on the same kernels, buffers layout and fused launches as inside the
    full network (entry 1x1 group, k x k branches, joint BatchNorm passes, packed filters, weight-gradient arena + unpack):
    block-level parity tests and single-block profiling.

        y  = forward(x_raw)      x_raw: the producer's PRE-activation values; the block reads relu(x_raw) (`relu_in`)
                                 y: the block's activated output relu(x_hat) (N, Co, H, W)
        dx = backward(dy)        dy = d loss / d y;  dx = d loss / d relu(x_raw);  parameter gradients are accumulated
                                 into p.grad like the full engine does
    """

    def __init__(self, eng: "HourglassEngine", mod: HG.Inception, N: int, H: int, W: int, relu_in: bool = True):
        self.eng, self.mod = eng, mod
        c_in, _ = HG.INCEPTION[mod.kind]
        Act.registry = []
        plan = {"steps": [], "convs": [], "stats_used": 0,
                "stats_arena": torch.zeros(4096 * L.STAT_SLOTS, 2, dtype=torch.float64, device=eng.device)}
        plan["x"] = eng._new(N, 3, H, W)       # only its shape is read (arena carving)
        self.x = Act(eng._new(N, c_in, H, W), 0, c_in, relu=relu_in)
        self.out = eng._inception(plan, plan["steps"], mod, self.x, N, H, W)
        plan["acts"], Act.registry = Act.registry, None
        eng._carve_arenas(plan)
        self.plan, self.step = plan, plan["steps"][0]

    @torch.no_grad()
    def forward(self, x_raw: torch.Tensor, training: bool = True) -> torch.Tensor:
        self.x.buf.copy_(x_raw)
        _native.zero_(self.plan["stats_arena"])
        self.eng._pack.run()
        self.eng._bias_table(self.plan).run()
        self.eng._run_forward(self.plan["steps"], training)
        o = self.out      # the buffer holds the raw convolution output: apply the BatchNorm like every consumer does
        v = o.buf[:, o.coff:o.coff + o.C]
        return torch.relu(v if o.scale is None else torch.addcmul(o.shift.view(1, -1, 1, 1), v, o.scale.view(1, -1, 1, 1)))

    @torch.no_grad()
    def backward(self, dy: torch.Tensor) -> torch.Tensor:
        o = self.out
        self.step.Pg[:, o.coff:o.coff + o.C].copy_(dy)
        _native.zero_(self.plan["sums_arena"])
        for a in self.plan["acts"]:
            a.grad_written = False
        self.eng._run_backward(self.plan["steps"])
        self.eng._finish_wgrads(self.plan)
        return self.x.gbuf.clone()


class _EngineFn(torch.autograd.Function):
    """One autograd node for the whole network; parameter gradients are written by the engine itself."""

    @staticmethod
    def forward(ctx, anchor, engine, x):
        ctx.engine = engine
        return engine._forward(x, need_grad=True).clone()

    @staticmethod
    def backward(ctx, dpred):
        ctx.engine._backward(dpred.contiguous())
        return None, None, None
