"""TEST INFRASTRUCTURE ONLY -- builds tests/emul/libsweep_emul.so (g++), the sequential host execution of the row-sweep
loss kernel's phase functions (see sweep_emul.cpp), and wraps it for numpy.  Used by tests/test_sweep_*_cpu.py."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "consistent_depth_amd", "csrc")
SO = os.path.join(HERE, "libsweep_emul.so")
_LIB = None


def build(force: bool = False) -> str:
    deps = [os.path.join(HERE, "sweep_emul.cpp"), os.path.join(CSRC, "loss_sweep_core.h"), os.path.join(CSRC, "loss_math.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", CSRC,
                               "-o", SO, deps[0]])
    return SO


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


GEO_FIELDS = ("H", "W", "PXT", "CG", "RP", "G", "RW", "R", "NG", "SMAX", "max_items", "ok")


def geo(H, W, pxt=2, ring_rows=0):
    out = (ctypes.c_int * 16)()
    ok = lib().sweep_emul_geo(H, W, pxt, ring_rows, out)
    g = dict(zip(GEO_FIELDS, list(out)))
    g["_raw"] = out
    return g if ok else None


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def plan(g, flow_fwd, flow_bwd, mask_fwd, mask_bwd):
    """Plan of ONE pair: items (n,4) int16 [p0, p1, w0, w1], lo/hi (2, NG)."""
    c = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    items = np.zeros((g["max_items"], 4), np.int16)
    lo, hi = np.zeros((2, g["NG"]), np.int16), np.zeros((2, g["NG"]), np.int16)
    ff, fb, mf, mb = c(flow_fwd), c(flow_bwd), c(mask_fwd), c(mask_bwd)
    n = lib().sweep_emul_plan(g["_raw"], _p(ff), _p(fb), _p(mf), _p(mb), _p(items), _p(lo), _p(hi))
    assert n > 0, n
    return items[:n], lo, hi


class FanInTooHigh(RuntimeError):
    pass


def fan_in(g, flow_fwd, flow_bwd, mask_fwd, mask_bwd):
    """Max number of valid sources that add to one target pixel of ONE pair (what the plan kernel counts)."""
    c = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    ff, fb, mf, mb = c(flow_fwd), c(flow_bwd), c(mask_fwd), c(mask_bwd)
    return lib().sweep_emul_fan_in(g["_raw"], _p(ff), _p(fb), _p(mf), _p(mb))


def loss(batch, lambda_r, lambda_b, mode=0, pxt=2, ring_rows=0, force_slow=False, order=0, service=False, fast=False, inw=True):
    c = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    depth = c(batch["depth"])
    B, _, H, W = depth.shape
    ff, fb, mf, mb = c(batch["flows"][0]), c(batch["flows"][1]), c(batch["masks"][0]), c(batch["masks"][1])
    intr, extr = c(batch["intrinsics"]), c(batch["extrinsics"])
    reproj, disp, total = np.zeros(B, np.float32), np.zeros(B, np.float32), np.zeros(1, np.float32)
    grad = np.zeros_like(depth)
    stats = (ctypes.c_long * 4)()
    lib().sweep_emul_set_order(int(order))
    lib().sweep_emul_set_service(int(service))
    lib().sweep_emul_set_fast(int(fast))
    lib().sweep_emul_set_inw(int(inw))
    fn = lib().sweep_emul_loss
    fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_float, ctypes.c_float] + [ctypes.c_int] * 7 + [ctypes.c_void_p] * 5
    rc = fn(_p(depth), _p(ff), _p(fb), _p(mf), _p(mb), _p(intr), _p(extr), lambda_r, lambda_b, mode, B, H, W, pxt, ring_rows,
            int(force_slow), _p(reproj), _p(disp), _p(total), _p(grad), stats)
    if rc == -11:
        raise FanInTooHigh("a target pixel receives more sources than a 32-bit accumulator can take: the product gives such a pair no plan "
                           "and recomputes it on the exact fallback path")
    if rc != 0:
        raise RuntimeError(f"sweep emulation failed: rc={rc}")
    return {"total": total, "reprojection": reproj, "disparity": disp, "grad_depth": grad,
            "slow_lanes": stats[0], "overflow_entries": stats[1], "items": stats[2], "degenerate": bool(stats[3])}


def inw_flags(g, flow_fwd, flow_bwd, mask_fwd, mask_bwd):
    """Rec::inw of every item of ONE pair's plan: (n_items, 2) int32 (1 = every valid tap of the item's group is a usable ring row)."""
    c = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    out = np.zeros((g["max_items"], 2), np.int32)
    ff, fb, mf, mb = c(flow_fwd), c(flow_bwd), c(mask_fwd), c(mask_bwd)
    n = lib().sweep_emul_inw(g["_raw"], _p(ff), _p(fb), _p(mf), _p(mb), _p(out))
    assert n > 0, n
    return out[:n]
