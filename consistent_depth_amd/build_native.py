"""Build the gfx950 C-ABI library (libcd_amd.so) in-tree with hipcc.

    python -m consistent_depth_amd.build_native [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  The library links only libamdhip64 (by soname), so inside a
PyTorch-ROCm process it binds to the HIP runtime torch already loaded.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
SO = os.path.join(PKG, "libcd_amd.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(PKG, "csrc", "*.h")) + glob.glob(os.path.join(REPO, "include", "*.h"))


def is_stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(s) > t for s in _deps())


def hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the native library cannot be built")


# Per-file flags.  loss_sweep.hip: the SLP vectorizer pairs the two pixels of a lane into v_pk_*_f32 instructions -- which issue at
# HALF the rate of their scalar forms on gfx950 (no gain) and need register moves to form the pairs: 80 v_mov per two items in the fast
# source pass, 412 vector instructions instead of 386 (static census, profiles/loss_sweep_isa_r05.txt).
EXTRA_FLAGS = {"loss_sweep.hip": ["-fno-slp-vectorize"]}


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO
    objdir = os.path.join(PKG, "csrc", "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-I", os.path.join(REPO, "include"), "-I", os.path.join(PKG, "csrc")]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdr_t = max(os.path.getmtime(h) for h in _deps() if h.endswith(".h"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc(), *flags, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
