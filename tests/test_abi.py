"""CPU: the C-ABI library loads and exports every symbol include/*.h declares; the Python
binding declares exactly the same set; the product path refuses to run without the GPU."""
import ctypes
import glob
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(cd_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from consistent_depth_amd import _native, build_native
    build_native.build()
    lib = ctypes.CDLL(_native.SO_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert declared == set(_native.SIGNATURES), "python binding and header disagree"


def test_abi_version_and_build_info():
    from consistent_depth_amd import _native
    lib = _native.lib()
    assert lib.cd_abi_version() == _native.ABI_VERSION
    assert b"gfx950" in lib.cd_build_info()
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "consistent_depth_amd.h")).read()
    assert int(re.search(r"#define CD_BN_STAT_SLOTS (\d+)", hdr).group(1)) == _native.BN_STAT_SLOTS


def test_workspace_query_is_pure_host():
    from consistent_depth_amd import _native
    lib = _native.lib()
    a = lib.cd_consistency_loss_workspace_bytes(4, 384, 224)
    b = lib.cd_consistency_loss_workspace_bytes(8, 384, 224)
    assert 0 < a < b < (1 << 28)   # B=8: ~17 MB (slabs of the evaluate-once gradient kernel dominate)
    assert lib.cd_consistency_loss_workspace_bytes(0, 384, 224) == 0


def test_product_path_has_no_cpu_fallback():
    import torch
    from consistent_depth_amd import synthetic
    from consistent_depth_amd.loss import consistency_loss as CL
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    b = synthetic.make_pair_batch(1, 16, 16, seed=0)
    t = torch.tensor
    with pytest.raises(RuntimeError, match="HIP device"):
        CL.consistency_loss(t(b["depth"]), [t(f) for f in b["flows"]], [t(m) for m in b["masks"]],
                            t(b["intrinsics"]), t(b["extrinsics"]), 1.0, 0.1)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for path in glob.glob(os.path.join(REPO, "consistent_depth_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
        assert "libcd_oracle" not in src, path
