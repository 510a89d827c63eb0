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


def test_weight_gradient_workspace_covers_every_plan():
    """Host-side consistency of the weight-gradient layouts (no GPU work): for every arithmetic mode and image size the packed
    partial sums of cd_conv2d_wgrad_plan -- splits x co groups x ci groups x taps x cob x cib floats -- fit the workspace that
    cd_conv2d_wgrad_workspace_floats sizes WITHOUT knowing the mode or the image (a caller allocates once)."""
    import ctypes
    from consistent_depth_amd import _native
    lib = _native.lib()
    before = lib.cd_get_conv_arith()
    shapes = [(16, 64, 11), (32, 64, 11), (64, 64, 11), (16, 64, 7), (32, 32, 7), (64, 32, 5), (32, 64, 3), (1, 64, 3), (128, 3, 7),
              (208, 128, 1), (224, 128, 1), (128, 128, 1), (256, 256, 1), (160, 256, 1), (112, 128, 1), (64, 128, 1), (32, 256, 1),
              (100, 40, 1), (24, 40, 11), (2048, 2048, 1), (8, 8, 3)]
    images = [(8, 384, 224), (8, 192, 112), (8, 96, 56), (8, 48, 28), (8, 24, 14), (2, 64, 96), (1, 13, 7), (2, 128, 512), (32, 384, 384)]
    try:
        for mode in (0, 1, 2):
            assert lib.cd_set_conv_arith(mode) == 0 and lib.cd_get_conv_arith() == mode
            for Cout, Cin, ks in shapes:
                ws = lib.cd_conv2d_wgrad_workspace_floats(Cout, Cin, ks)
                assert ws > 0
                for N, H, W in images:
                    cob, cib, splits = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    assert lib.cd_conv2d_wgrad_plan(Cout, Cin, ks, N, H, W, ctypes.byref(cob), ctypes.byref(cib), ctypes.byref(splits)) == 0
                    assert cob.value > 0 and cib.value > 0 and splits.value >= 1
                    need = (-(-Cout // cob.value)) * (-(-Cin // cib.value)) * ks * ks * cob.value * cib.value * splits.value
                    assert need <= ws, (mode, Cout, Cin, ks, N, H, W, cob.value, cib.value, splits.value, need, ws)
        assert lib.cd_set_conv_arith(3) != 0     # unknown mode
    finally:
        lib.cd_set_conv_arith(before)


def test_packed_filter_holds_every_layout():
    """cd_conv2d_packed_weight_floats does not depend on the arithmetic mode (the packed buffer carries the fp32 layout and, where a
    split kernel exists, its bf16 layout), so a mode switch never needs a re-allocation."""
    from consistent_depth_amd import _native
    lib = _native.lib()
    before = lib.cd_get_conv_arith()
    try:
        sizes = {}
        for mode in (0, 1, 2):
            lib.cd_set_conv_arith(mode)
            sizes[mode] = [lib.cd_conv2d_packed_weight_floats(co, ci, ks, tr) for co, ci, ks in ((16, 64, 11), (208, 128, 1), (64, 32, 3), (128, 3, 7), (1, 64, 3))
                           for tr in (0, 1)]
        assert sizes[0] == sizes[1] == sizes[2] and all(v > 0 for v in sizes[0])
    finally:
        lib.cd_set_conv_arith(before)


def test_host_record_layouts_match_the_public_header(tmp_path):
    """Every struct the Python host fills byte by byte (numpy record types, ctypes.Structure) against the C layout of
    include/consistent_depth_amd.h -- compiled with gcc as plain C (the header needs no HIP): size and the offset of EVERY field.
    (Round 4 shipped a 124-byte record for a 128-byte struct to the GPU box once; this catches that on the CPU.)"""
    import ctypes
    import re
    import subprocess
    import numpy as np
    from consistent_depth_amd.loaders import pair_store
    from consistent_depth_amd.ops import conv
    hdr = open(os.path.join(REPO, "include", "consistent_depth_amd.h")).read()

    def c_fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.sub(r"\[.*\]", "", first.split()[-1].lstrip("*")))
            names += [re.sub(r"\[.*\]", "", r.strip().lstrip("*")) for r in rest]
        return names

    records = {
        "cd_conv_desc": np.dtype(conv._CONV_DESC_DT),
        "cd_wgrad_desc": np.dtype(conv.WgradTable._DT),
        "cd_unpack_desc": np.dtype(conv.UnpackTable._DT),
        "cd_pack_desc": np.dtype(conv.PackTable._DT),
    }
    structs = {"cd_pair_store": pair_store._StoreDesc, "cd_pair_batch": pair_store._BatchDesc}
    src = ["#include <stdio.h>", "#include <stddef.h>", "#include <stdint.h>", '#include "consistent_depth_amd.h"', "int main(void) {"]
    for s in list(records) + list(structs):
        src.append(f'printf("{s} size %zu\\n", sizeof({s}));')
        for f in c_fields(s):
            src.append(f'printf("{s} {f} %zu\\n", offsetof({s}, {f}));')
    src += ["return 0; }"]
    (tmp_path / "layout.c").write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(tmp_path / "layout.c"), "-o", str(tmp_path / "layout")])
    out = subprocess.check_output([str(tmp_path / "layout")], text=True)
    c = {}
    for line in out.splitlines():
        s, f, v = line.split()
        c.setdefault(s, []).append((f, int(v)))
    for s, dt in records.items():
        assert dt.itemsize == dict(c[s])["size"], (s, dt.itemsize, dict(c[s])["size"])
        offs = [v for f, v in c[s] if f != "size"]
        mine = [dt.fields[n][1] for n in dt.names]
        # one numpy field per C field, in order, at the same offsets; a C array field (cd_wgrad_desc.pad[2]) is several consecutive
        # numpy fields: every C offset must be a numpy offset, and the numpy fields in between must be 4-byte ints that tile the gap
        assert offs == sorted(offs) and mine == sorted(mine) and set(offs) <= set(mine), (s, offs, mine)
        assert [o for o in mine if o in set(offs)] == offs, (s, offs, mine)
        for a, b in zip(mine, mine[1:] + [dt.itemsize]):
            assert b - a == dt.fields[dt.names[mine.index(a)]][0].itemsize, (s, a, b)       # no holes, no overlaps
    for s, st in structs.items():
        assert ctypes.sizeof(st) == dict(c[s])["size"], s
        assert [getattr(st, n).offset for n, _ in st._fields_] == [v for f, v in c[s] if f != "size"], s
