"""Thin Python face of the MFMA convolution entry points (C ABI: cd_conv2d_*).
Tensors are NCHW fp32 on the HIP device; `(tensor, channel offset)` pairs address channel slices
of concat buffers in place."""
from __future__ import annotations

import os

import torch

from .. import _native

KERNEL_SIZES = (1, 3, 5, 7, 11)


def pack_weights(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """w (Cout, Cin, k, k) -> packed filter (forward, or the flipped/transposed dgrad filter)."""
    Cout, Cin, k, k2 = w.shape
    assert k == k2 and k in KERNEL_SIZES, f"kernel size {k} not supported"
    lib = _native.lib()
    n = lib.cd_conv2d_packed_weight_floats(Cout, Cin, k, int(transposed))
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    wc = w.detach().contiguous()
    rc = lib.cd_conv2d_pack_weights(_native.dev_ptr(wc, "weight"), Cout, Cin, k, int(transposed), out.data_ptr(),
                                    _native.stream_ptr(w.device))
    _native.check(rc, "cd_conv2d_pack_weights")
    return out


def conv2d(x, packed_w, Cin, Cout, ks, bias=None, x_coff=0, out=None, y_coff=0, in_scale=None, in_shift=None,
           in_relu=False, stats=None, accumulate=False, cfg=None):
    """out[:, y_coff:y_coff+Cout] = conv(act(x[:, x_coff:x_coff+Cin])) + bias.  Returns `out`.
    cfg = (tile_rows, co_tiles) launch shape (see `tuned_config`); None = the library's heuristic."""
    N, x_ctot, H, W = x.shape
    if out is None:
        out = torch.empty(N, Cout, H, W, dtype=torch.float32, device=x.device)
    y_ctot = out.shape[1]
    opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else None  # noqa: E731
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.is_cuda and stats.is_contiguous() and \
            stats.numel() == _native.BN_STAT_SLOTS * 2 * y_ctot, "stats must be (BN_STAT_SLOTS, y_ctot, 2) fp64"
    ty, cot = cfg if cfg is not None else (0, 0)
    rc = _native.lib().cd_conv2d_fwd_cfg(
        _native.dev_ptr(x, "x"), x_ctot, x_coff, Cin, _native.dev_ptr(packed_w, "packed_w"), opt(bias, "bias"),
        opt(in_scale, "in_scale"), opt(in_shift, "in_shift"), int(in_relu), _native.dev_ptr(out, "out"), y_ctot, y_coff,
        Cout, stats.data_ptr() if stats is not None else None, int(accumulate), N, H, W, ks, ty, cot,
        _native.stream_ptr(x.device))
    _native.check(rc, "cd_conv2d_fwd_cfg")
    return out


_TUNED = {}
_TUNE_CACHE_LOADED = [False]


def autotune_enabled() -> bool:
    return os.environ.get("CD_AMD_CONV_AUTOTUNE", "1") != "0"


def _tune_cache_path():
    return os.environ.get("CD_AMD_CONV_TUNE_CACHE") or None


def _load_tune_cache():
    """CD_AMD_CONV_TUNE_CACHE=<file>: measured launch shapes persist across processes (JSON, key -> [rows, tiles])."""
    if _TUNE_CACHE_LOADED[0]:
        return
    _TUNE_CACHE_LOADED[0] = True
    path = _tune_cache_path()
    if path and os.path.exists(path):
        import json
        try:
            with open(path) as f:
                for k, v in json.load(f).items():
                    _TUNED[tuple(json.loads(k))] = tuple(v) if v is not None else None
        except (OSError, ValueError):
            pass   # unreadable cache: measure again


def _save_tune_cache():
    path = _tune_cache_path()
    if not path:
        return
    import json
    tmp = f"{path}.{os.getpid()}.tmp"
    try:
        with open(tmp, "w") as f:
            json.dump({json.dumps([x if isinstance(x, str) else (bool(x) if isinstance(x, bool) else int(x)) for x in k]): v for k, v in _TUNED.items()}, f)
        os.replace(tmp, path)
    except OSError:
        pass


# launch-shape hints the tuners try (CD_AMD_CONV_SHAPE32=0: without round 6's 8-tile / two-chunk shape -- A/B measurements)
_TILE_HINTS = (4, 8, 16, 32) if os.environ.get("CD_AMD_CONV_SHAPE32", "1") != "0" else (4, 8, 16)


def tuned_config(ks, Cin, Cout, N, H, W, device, *, affine_in=False, relu_in=False, stats=False, accumulate=False,
                 x_ctot=None, y_ctot=None, iters=3):
    """(tile_rows, co_tiles) of the fastest launch shape for this convolution, measured once per distinct
    (shape, fusion flags) on scratch tensors with HIP events and cached for the life of the process -- the
    result of the convolution does not depend on the choice.  Returns None (library heuristic) when disabled."""
    if not autotune_enabled():
        return None
    x_ctot, y_ctot = x_ctot or Cin, y_ctot or Cout
    _load_tune_cache()
    arith = _native.lib().cd_get_conv_arith()     # the arithmetic modes are different kernels
    key = (ks, Cin, Cout, N, H, W, int(bool(affine_in)), int(bool(relu_in)), int(bool(stats)), int(bool(accumulate)), x_ctot, y_ctot, arith)
    if key in _TUNED:
        return _TUNED[key]
    dev = torch.device(device)
    x = torch.randn(N, x_ctot, H, W, device=dev)
    out = torch.zeros(N, y_ctot, H, W, device=dev)
    pk = torch.randn(_native.lib().cd_conv2d_packed_weight_floats(Cout, Cin, ks, 0), device=dev) * 0.05
    sc = torch.rand(Cin, device=dev) + 0.5 if affine_in else None
    sh = torch.randn(Cin, device=dev) * 0.1 if affine_in else None
    st = torch.zeros(_native.BN_STAT_SLOTS, y_ctot, 2, dtype=torch.float64, device=dev) if stats else None
    max_cot = _native.lib().cd_conv2d_packed_co_tiles(Cout, ks)
    best, best_t = None, float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for ty in _TILE_HINTS:        # (32: 8 row tiles + two channel chunks per round, split-bf16 k x k kernels only -- refused elsewhere)
        for cot in (1, 2, 4, 8, 16):
            if cot > max_cot or (cot == 16 and ty > 4) or (cot == 8 and ty > 8) or (ty == 32 and cot > 1):
                continue

            def run():
                conv2d(x, pk, Cin, Cout, ks, out=out, in_scale=sc, in_shift=sh, in_relu=relu_in, stats=st,
                       accumulate=accumulate, cfg=(ty, cot))
            try:
                run()
            except RuntimeError:      # launch shape not available for this filter (LDS budget)
                continue
            t = float("inf")
            for _ in range(2):        # best of two repetitions: one noisy sample must not pick the launch shape
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                e1.synchronize()
                t = min(t, e0.elapsed_time(e1))
            if t < best_t:
                best, best_t = (ty, cot), t
    _TUNED[key] = best
    _save_tune_cache()
    return best


_CONV_DESC_DT = [("x", "<u8"), ("packed_w", "<u8"), ("bias", "<u8"), ("in_scale", "<u8"), ("in_shift", "<u8"), ("y", "<u8"), ("stats", "<u8"),
                 ("x_ctot", "<i4"), ("x_coff", "<i4"), ("Cin", "<i4"), ("in_relu", "<i4"), ("y_ctot", "<i4"), ("y_coff", "<i4"), ("Cout", "<i4"),
                 ("accumulate", "<i4"), ("N", "<i4"), ("H", "<i4"), ("W", "<i4"), ("ks", "<i4")]


def conv2d_multi(members, cfg=None) -> bool:
    """SEVERAL convolutions of one launch shape in ONE dispatch (cd_conv2d_fwd_multi): `members` = up to 4 dicts with the arguments of
    conv2d (x, packed_w, Cin, Cout, ks, bias, x_coff, out, y_coff, in_scale, in_shift, in_relu, stats, accumulate), same N, H, W and
    Cout, largest filter first.  Returns False -- nothing launched -- when the library has no such dispatch for them (the fp32
    arithmetic mode): the caller then launches the members one by one.  Same bits either way."""
    import ctypes
    import numpy as np
    tab = np.zeros(len(members), np.dtype(_CONV_DESC_DT))
    opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else 0  # noqa: E731
    dev = None
    for i, m in enumerate(members):
        x, out = m["x"], m["out"]
        N, x_ctot, H, W = x.shape
        st = m.get("stats")
        if st is not None:
            assert st.dtype == torch.float64 and st.is_cuda and st.is_contiguous() and st.numel() == _native.BN_STAT_SLOTS * 2 * out.shape[1]
        tab[i] = (_native.dev_ptr(x, "x"), _native.dev_ptr(m["packed_w"], "packed_w"), opt(m.get("bias"), "bias"), opt(m.get("in_scale"), "in_scale"),
                  opt(m.get("in_shift"), "in_shift"), _native.dev_ptr(out, "out"), st.data_ptr() if st is not None else 0,
                  x_ctot, m.get("x_coff", 0), m["Cin"], int(bool(m.get("in_relu", False))), out.shape[1], m.get("y_coff", 0), m["Cout"],
                  int(bool(m.get("accumulate", False))), N, H, W, m["ks"])
        dev = x.device
    ty, cot = cfg if cfg is not None else (0, 0)
    rc = _native.lib().cd_conv2d_fwd_multi(tab.ctypes.data_as(ctypes.c_void_p), len(members), ty, cot, _native.stream_ptr(dev))
    if rc == -4:      # CD_ERR_UNSUPPORTED
        return False
    _native.check(rc, "cd_conv2d_fwd_multi")
    return True


def tuned_multi(members, single_cfgs, iters=3):
    """Is ONE dispatch of `members` (conv2d_multi) faster than their own launches with their timed shapes `single_cfgs`, and with
    which shared launch shape?  Timed once per distinct (shapes, fusion flags) on the members' own buffers with HIP events -- the
    results do not depend on the choice -- and cached for the life of the process.  Returns the (tile_rows, co_tiles) of the
    fastest merged launch, or None: launch them one by one."""
    if not autotune_enabled() or os.environ.get("CD_AMD_CONV_MULTI", "1") == "0":
        return None
    _load_tune_cache()
    arith = _native.lib().cd_get_conv_arith()
    x0 = members[0]["x"]
    N, _, H, W = x0.shape
    key = ("multi", N, H, W, arith) + tuple(v for m in members for v in (m["ks"], m["Cin"], m["Cout"], m["x"].shape[1], m["out"].shape[1],
                                                                         int(m.get("in_scale") is not None), int(bool(m.get("in_relu", False))),
                                                                         int(m.get("stats") is not None), int(bool(m.get("accumulate", False)))))
    if key in _TUNED:
        return _TUNED[key]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        t = float("inf")
        for _ in range(2):
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            t = min(t, e0.elapsed_time(e1))
        return t

    def singles():
        for m, cfg in zip(members, single_cfgs):
            conv2d(m["x"], m["packed_w"], m["Cin"], m["Cout"], m["ks"], bias=m.get("bias"), x_coff=m.get("x_coff", 0), out=m["out"],
                   y_coff=m.get("y_coff", 0), in_scale=m.get("in_scale"), in_shift=m.get("in_shift"), in_relu=m.get("in_relu", False),
                   stats=m.get("stats"), accumulate=m.get("accumulate", False), cfg=cfg)
    # a merged launch must clearly win against the members' own launches (3 %); among merged shapes the fastest one is kept
    # (round 4 compared every later shape with 0.97 x the best MERGED time: a shape 1-2 % faster than the accepted one was rejected)
    single_t = timed(singles)
    best, best_t = None, float("inf")
    for ty in _TILE_HINTS:
        for cot in (1, 2):
            if ty == 32 and cot > 1:
                continue
            if not conv2d_multi(members, (ty, cot)):
                if ty == 32:        # (that shape exists for 32 output channels per column tile only)
                    continue
                _TUNED[key] = None
                return None
            t = timed(lambda: conv2d_multi(members, (ty, cot)))
            if t < 0.97 * single_t and t < best_t:
                best, best_t = (ty, cot), t
    _TUNED[key] = best
    _save_tune_cache()
    return best


def wgrad_workspace_floats(Cout, Cin, ks) -> int:
    return _native.lib().cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks)


def wgrad_workspace(Cout, Cin, ks, device) -> torch.Tensor:
    return torch.empty(wgrad_workspace_floats(Cout, Cin, ks), dtype=torch.float32, device=device)


def conv2d_wgrad(x, dy, Cin, Cout, ks, dw, workspace, x_coff=0, dy_coff=0, in_scale=None, in_shift=None,
                 in_relu=False, accumulate=False, prezeroed=False):
    """dw (Cout,Cin,ks,ks) (+)= sum dy[:, dy_coff:+Cout] * act(x[:, x_coff:+Cin]) shifted by the taps.
    dw=None: deferred form -- the result stays packed in `workspace` for an UnpackTable."""
    N, x_ctot, H, W = x.shape
    opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else None  # noqa: E731
    flags = int(accumulate) | (2 if prezeroed else 0) | (4 if dw is None else 0)
    rc = _native.lib().cd_conv2d_wgrad(
        _native.dev_ptr(x, "x"), x_ctot, x_coff, Cin, opt(in_scale, "in_scale"), opt(in_shift, "in_shift"), int(in_relu),
        _native.dev_ptr(dy, "dy"), dy.shape[1], dy_coff, Cout, opt(dw, "dw"), flags,
        _native.dev_ptr(workspace, "workspace"), N, H, W, ks, _native.stream_ptr(x.device))
    _native.check(rc, "cd_conv2d_wgrad")
    return dw


def wgrad_plan(Cout, Cin, ks, N, H, W):
    """(cob, cib, splits): channel block sizes and number of per-workgroup slices of the packed layout cd_conv2d_wgrad
    uses for these arguments."""
    import ctypes
    cob, cib, splits = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _native.check(_native.lib().cd_conv2d_wgrad_plan(Cout, Cin, ks, N, H, W, ctypes.byref(cob), ctypes.byref(cib), ctypes.byref(splits)),
                  "cd_conv2d_wgrad_plan")
    return cob.value, cib.value, splits.value


class WgradTable:
    """MANY weight gradients in a handful of launches (cd_conv2d_wgrad_desc / cd_conv2d_wgrad_table): one launch per kernel class
    (filter size x channel-block shape) for every gradient of a network.  add() registers a gradient in deferred form (the
    arguments of conv2d_wgrad(dw=None)) and returns False when it is not one the table kernels compute (1x1, the RGB stem, the fp32
    arithmetic mode): the caller launches that one itself.  All operand tensors are static buffers of a plan, so the device tables
    are built once (build()); run() enqueues the launches.  Results are bit-identical to one conv2d_wgrad per gradient."""

    _DT = [("x", "<u8"), ("in_scale", "<u8"), ("in_shift", "<u8"), ("dy", "<u8"), ("workspace", "<u8"),
           ("x_ctot", "<i4"), ("x_coff", "<i4"), ("Cin", "<i4"), ("in_relu", "<i4"), ("dy_ctot", "<i4"), ("dy_coff", "<i4"), ("Cout", "<i4"),
           ("N", "<i4"), ("H", "<i4"), ("W", "<i4"), ("ks", "<i4"),
           ("klass", "<i4"), ("splits", "<i4"), ("cigs", "<i4"), ("zpg", "<i4"), ("cogs", "<i4"), ("tiles_x", "<i4"), ("tiles_y", "<i4"),
           ("blocks", "<i4"), ("block_end", "<i4"), ("pad0", "<i4"), ("pad1", "<i4")]
    MAX_PER_LAUNCH = 64

    def __init__(self, device):
        import numpy as np
        self.device, self._descs, self._keep, self._launches = device, [], [], None
        assert np.dtype(self._DT).itemsize == 128

    def add(self, x, dy, Cin, Cout, ks, workspace, x_coff=0, dy_coff=0, in_scale=None, in_shift=None, in_relu=False) -> bool:
        import ctypes
        import numpy as np
        N, x_ctot, H, W = x.shape
        d = np.zeros(1, np.dtype(self._DT))
        opt = lambda t, name: _native.dev_ptr(t, name) if t is not None else 0  # noqa: E731
        d[0] = (_native.dev_ptr(x, "x"), opt(in_scale, "in_scale"), opt(in_shift, "in_shift"), _native.dev_ptr(dy, "dy"),
                _native.dev_ptr(workspace, "workspace"), x_ctot, x_coff, Cin, int(in_relu), dy.shape[1], dy_coff, Cout, N, H, W, ks) + (0,) * 11
        rc = _native.lib().cd_conv2d_wgrad_desc(d.ctypes.data_as(ctypes.c_void_p))
        _native.check(rc, "cd_conv2d_wgrad_desc")
        if int(d[0]["klass"]) < 0:
            return False
        self._descs.append(d)
        self._keep += [x, dy, workspace, in_scale, in_shift]      # the table holds raw pointers
        self._launches = None
        return True

    def __len__(self):
        return len(self._descs)

    def build(self):
        import numpy as np
        self._launches = []
        by_class = {}
        for d in self._descs:
            by_class.setdefault(int(d[0]["klass"]), []).append(d)
        for klass, ds in sorted(by_class.items()):
            # heaviest first: workgroups are dispatched in table order, the long ones must not start last.  Work of one workgroup ~
            # (image tiles it walks) x taps; ties broken by the registration order (deterministic tables)
            def weight(d):
                items = int(d[0]["N"]) * int(d[0]["tiles_x"]) * int(d[0]["tiles_y"])
                return -((items + int(d[0]["splits"]) - 1) // int(d[0]["splits"])) * int(d[0]["ks"]) ** 2
            ds = sorted(ds, key=weight)
            for s0 in range(0, len(ds), self.MAX_PER_LAUNCH):
                part = np.concatenate(ds[s0:s0 + self.MAX_PER_LAUNCH])
                part["block_end"] = np.cumsum(part["blocks"])
                tab = torch.from_numpy(part.view(np.uint8).copy()).to(self.device)
                self._launches.append((klass, tab, len(part), int(part["block_end"][-1])))
        return self

    def run(self):
        if not self._descs:
            return
        if self._launches is None:
            self.build()
        lib, stream = _native.lib(), _native.stream_ptr(self.device)
        for klass, tab, n, total in self._launches:
            _native.check(lib.cd_conv2d_wgrad_table(tab.data_ptr(), n, klass, total, stream), "cd_conv2d_wgrad_table")


class UnpackTable:
    """Deferred weight gradients of a whole network written by ONE launch (cd_conv2d_wgrad_unpack_table).

    add(workspace, grad, Cin, ks, plan, row0=0) registers a destination gradient tensor (rows = grad.shape[0]) fed from the
    output-channel rows [row0, row0+rows) of a packed workspace; run() launches, rebuilding the device table first if a
    gradient tensor moved (FlatAdam re-homes .grad; zero_grad(set_to_none) would reallocate).
    The gradients are ACCUMULATED into their destinations (p.grad += dW, torch's convention): whatever another autograd
    node has already put there -- the lambda_parameter * sign(p - p0) pull of ParameterLoss -- survives; zero_grad()
    clears them once per step."""

    _DT = [("packed", "<u8"), ("dw", "<u8"), ("Cin", "<i4"), ("ks", "<i4"), ("cob", "<i4"), ("cib", "<i4"), ("cig", "<i4"),
           ("row0", "<i4"), ("rows", "<i4"), ("acc", "<i4"), ("splits", "<i4"), ("split_stride", "<i4")]

    def __init__(self, device):
        self.device, self._entries, self._table, self._ptrs = device, [], None, None

    def add(self, workspace, grad_of, Cin, ks, plan, row0=0, cout_total=None):
        """grad_of: callable returning the CURRENT gradient tensor (evaluated at every run); plan = wgrad_plan(...) of the
        convolution that filled `workspace`, cout_total its output channels (default: the destination's rows)."""
        self._entries.append((workspace, grad_of, Cin, ks, plan, row0, cout_total))

    def _refresh(self, grads):
        import numpy as np
        tab = np.zeros(len(self._entries), np.dtype(self._DT))
        for j, ((ws, _, Cin, ks, (cob, cib, splits), row0, cout_total), g) in enumerate(zip(self._entries, grads)):
            if not (g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda and g.shape[1] == Cin and g.shape[2] == ks):
                raise RuntimeError("weight gradients must be contiguous fp32 (rows, Cin, k, k) on the HIP device")
            cout = cout_total if cout_total is not None else g.shape[0]
            cig = (Cin + cib - 1) // cib
            stride = ((cout + cob - 1) // cob) * cig * ks * ks * cob * cib      # one workgroup slice of the packed buffer
            if stride * splits > ws.numel():
                raise RuntimeError("wgrad workspace smaller than its plan")
            tab[j] = (ws.data_ptr(), g.data_ptr(), Cin, ks, cob, cib, cig, row0, g.shape[0], 1, splits, stride)
        self._ptrs = [g.data_ptr() for g in grads]
        self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)

    def run(self):
        if not self._entries:
            return
        grads = [e[1]() for e in self._entries]
        if self._ptrs is None or any(g.data_ptr() != q for g, q in zip(grads, self._ptrs)):
            self._refresh(grads)
        rc = _native.lib().cd_conv2d_wgrad_unpack_table(self._table.data_ptr(), len(self._entries), _native.stream_ptr(self.device))
        _native.check(rc, "cd_conv2d_wgrad_unpack_table")


class PackTable:
    """All filters of a network packed by ONE kernel launch (cd_conv2d_pack_weights_table).

    new_filter(OC, IC, ks) reserves a packed filter of the logical conv IC -> OC; source(filter, param,
    transposed, oc_off, ic_off) registers a parameter that fills part of it (several parameters may be
    concatenated into one fused filter); build() allocates one zeroed arena and the device descriptor table.
    Parameters may be re-homed later (FlatAdam moves them into its flat buffer): run() re-checks the source
    pointers and refreshes the table when they moved."""

    _DT = [("w", "<u8"), ("packed", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"), ("ks", "<i4"), ("tr", "<i4"),
           ("OC", "<i4"), ("IC", "<i4"), ("oc_off", "<i4"), ("ic_off", "<i4")]

    def __init__(self, device):
        self.device, self._filters, self._sources = device, [], []
        self._arena = self._table = self._ptrs = None
        self.views = []

    def new_filter(self, OC: int, IC: int, ks: int) -> int:
        n = _native.lib().cd_conv2d_packed_weight_floats(OC, IC, ks, 0)
        self._filters.append((OC, IC, ks, n))
        return len(self._filters) - 1

    def source(self, filt: int, param: torch.nn.Parameter, transposed: bool, oc_off: int = 0, ic_off: int = 0):
        self._sources.append((filt, param, bool(transposed), oc_off, ic_off))

    def add(self, param: torch.nn.Parameter, transposed: bool) -> int:
        """One parameter = one filter (forward form, or its dgrad twin)."""
        Cout, Cin, k, _ = param.shape
        f = self.new_filter(Cin if transposed else Cout, Cout if transposed else Cin, k)
        self.source(f, param, transposed)
        return f

    def build(self):
        total = sum((n + 63) // 64 * 64 for _, _, _, n in self._filters)
        self._arena = torch.zeros(total, dtype=torch.float32, device=self.device)  # padding stays zero forever
        self.views, off = [], 0
        for _, _, _, n in self._filters:
            self.views.append(self._arena[off:off + n])
            off += (n + 63) // 64 * 64
        self._refresh_table()
        return self

    def _refresh_table(self):
        import numpy as np
        tab = np.zeros(len(self._sources), np.dtype(self._DT))
        for j, (f, p, tr, oo, io) in enumerate(self._sources):
            if not (p.is_contiguous() and p.dtype == torch.float32 and p.is_cuda):
                raise RuntimeError("conv weights must be contiguous fp32 on the HIP device")
            OC, IC, ks, _ = self._filters[f]
            assert p.shape[2] == ks
            tab[j] = (p.data_ptr(), self.views[f].data_ptr(), p.shape[0], p.shape[1], ks, int(tr), OC, IC, oo, io)
        self._ptrs = [p.data_ptr() for _, p, _, _, _ in self._sources]
        self._table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)

    def view(self, index: int) -> torch.Tensor:
        return self.views[index]

    def run(self):
        if any(p.data_ptr() != q for (_, p, _, _, _), q in zip(self._sources, self._ptrs)):
            self._refresh_table()
        rc = _native.lib().cd_conv2d_pack_weights_table(self._table.data_ptr(), len(self._sources),
                                                        _native.stream_ptr(self.device))
        _native.check(rc, "cd_conv2d_pack_weights_table")
