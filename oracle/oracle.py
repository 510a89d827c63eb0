"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle.

Loads oracle/libcd_oracle.so (plain C restatement, see cd_oracle.c for the reference
file:line map and the parity-pinning statement) and exposes numpy-in / numpy-out
helpers.  May be imported ONLY by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; nothing under consistent_depth_amd/ imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcd_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cd_oracle.c", "cd_oracle_body.inc")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def consistency_loss(depth, flows, masks, intr, extr, lambda_r=1.0, lambda_b=0.1,
                     dtype=np.float64, want_grad=True):
    """Reference loss on numpy inputs.  Returns dict(reprojection, disparity, total, grad_depth)."""
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    sfx = "f64" if dtype == np.float64 else "f32"
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    depth, f0, f1, m0, m1 = c(depth), c(flows[0]), c(flows[1]), c(masks[0]), c(masks[1])
    intr, extr = c(intr), c(extr)
    B, _, H, W = depth.shape
    reproj, disp, total = np.zeros(B, dtype), np.zeros(B, dtype), np.zeros(1, dtype)
    grad = np.zeros_like(depth) if want_grad else None
    fn = getattr(lib(), f"cd_oracle_consistency_loss_{sfx}")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 7 + [ct, ct, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    rc = fn(_p(depth), _p(f0), _p(f1), _p(m0), _p(m1), _p(intr), _p(extr), ct(lambda_r), ct(lambda_b),
            B, H, W, _p(reproj), _p(disp), _p(total), _p(grad) if want_grad else None)
    assert rc == 0
    return {"reprojection": reproj, "disparity": disp, "total": total, "grad_depth": grad}


def sample(data, uv, dtype=np.float64):
    sfx = "f64" if dtype == np.float64 else "f32"
    data, uv = np.ascontiguousarray(data, dtype), np.ascontiguousarray(uv, dtype)
    B, C, H, W = data.shape
    out = np.zeros_like(data)
    fn = getattr(lib(), f"cd_oracle_sample_{sfx}")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    assert fn(_p(data), _p(uv), B, C, H, W, _p(out)) == 0
    return out


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, dtype=np.float32):
    """In-place Adam step on numpy arrays (step is 1-based)."""
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    sfx = "f64" if dtype == np.float64 else "f32"
    for a in (p, g, m, v):
        assert a.dtype == dtype and a.flags.c_contiguous
    fn = getattr(lib(), f"cd_oracle_adam_step_{sfx}")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ct, ct, ct, ct, ctypes.c_int]
    assert fn(_p(p), _p(g), _p(m), _p(v), p.size, ct(lr), ct(b1), ct(b2), ct(eps), int(step)) == 0


def rel_l1(a, b) -> float:
    """Relative L1 distance |a-b|_1 / |b|_1 (the parity measure BASELINE.json names)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.abs(b).sum()
    return float(np.abs(a - b).sum() / (den if den > 0 else 1.0))
