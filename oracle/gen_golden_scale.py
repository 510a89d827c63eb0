#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).  tests/golden/scale_stage_6f_48x40.npz = what the reference's
OWN lines scale_calibration.py:228-311 write for the seeded case of oracle/scale_oracle.py::make_case -- scales.csv, the keys of
metadata_scaled.npz, the scaled depth maps -- executed unmodified by oracle/scale_oracle.py::reference_stage.

    python -m oracle.gen_golden_scale
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SEED = 3
GOLDEN = os.path.join(REPO, "tests", "golden", "scale_stage_6f_48x40.npz")


def main():
    from oracle import scale_oracle as S
    inv_src, inv_cmp, intr, extr = S.make_case(SEED)
    tmp = tempfile.mkdtemp()
    path, out = os.path.join(tmp, "clip"), os.path.join(tmp, "out")
    S.write_case(path, out, inv_src, inv_cmp, intr, extr)
    ref = S.reference_stage(path, out, sorted(inv_src))
    res = {"seed": np.array(SEED), "scales_csv": ref["scales_csv"], "intrinsics": ref["meta"]["intrinsics"], "extrinsics": ref["meta"]["extrinsics"],
           "scales": ref["meta"]["scales"], "scaled_frames": np.array(sorted(ref["scaled"])),
           "scaled": np.stack([ref["scaled"][i] for i in sorted(ref["scaled"])])}
    np.savez_compressed(GOLDEN, **res)
    print("wrote", GOLDEN, os.path.getsize(GOLDEN), "bytes; valid frames", res["scaled_frames"].tolist(), "scales", res["scales_csv"][:, 1].tolist())


if __name__ == "__main__":
    main()
