#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the flow-consistency mask builder, produced by RUNNING THE REFERENCE
(/root/reference/utils/consistency.py, imported unmodified; build container only).

    python oracle/gen_golden_masks.py        # writes tests/golden/masks_*.npz
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from utils import consistency as ref_consistency  # noqa: E402  (the reference's module)


from oracle.gen_golden_masks_inputs import make_case  # noqa: E402


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    for name, H, W, seed, wild, ft, ct in (("masks_basic_24x40", 24, 40, 1, False, 1.0, 1.0), ("masks_tight_33x47", 33, 47, 2, False, 0.5, 0.25),
                                         ("masks_wild_32x32", 32, 32, 3, True, 1.0, 1.0), ("masks_size_96x128", 96, 128, 4, False, 1.0, 0.6)):
        flows, colors = make_case(H, W, seed, wild)
        masks = ref_consistency.consistent_flow_masks(flows, colors, ft, ct)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), flow_fwd=flows[0], flow_bwd=flows[1], color0=colors[0], color1=colors[1],
                            flow_thresh=np.float64(ft), color_thresh=np.float64(ct), mask_fwd=masks[0], mask_bwd=masks[1])
        print(name, "valid fraction", float(masks[0].mean()), float(masks[1].mean()))


if __name__ == "__main__":
    main()
