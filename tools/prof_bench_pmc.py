#!/usr/bin/env python3
"""Per-kernel-family PMC means from rocprofv3 --pmc runs of bench.py (rocpd sqlite): MFMA busy fraction etc."""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]


def fam(name):
    m = re.search(r"(conv_fwd_kernel|conv_wgrad_kernel)<(\d+)", name)
    if m:
        return f"{m.group(1)}<k={m.group(2)}>"
    m = re.search(r"cd::(\w+)", name)
    return m.group(1) if m else "other"


for db in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    per = defaultdict(float)
    names = {}
    for name, cn, cv, did in c.execute("select name, counter_name, counter_value, dispatch_id from pmc_events"):
        per[(did, cn)] += cv
        names[did] = fam(name)
    agg = defaultdict(lambda: defaultdict(list))
    for (did, cn), v in per.items():
        agg[names[did]][cn].append(v)
    print("==", os.path.relpath(db, root))
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        line = f"  {k:28s} n={len(next(iter(cs.values()))):5d} "
        tot = {cn: sum(v) for cn, v in cs.items()}
        line += " ".join(f"{cn}={tot[cn]:.3e}" for cn in sorted(tot))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CYCLES" in tot and tot["SQ_BUSY_CYCLES"]:
            # MFMA busy cycles are summed over SIMDs; SQ_BUSY_CYCLES over SEs (x4 SIMD... report the raw ratio)
            line += f"  mfma_busy/sq_busy={tot['SQ_VALU_MFMA_BUSY_CYCLES'] / tot['SQ_BUSY_CYCLES']:.3f}"
        if "GRBM_GUI_ACTIVE" in tot and "SQ_VALU_MFMA_BUSY_CYCLES" in tot and tot["GRBM_GUI_ACTIVE"]:
            # 1024 SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs
            util = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (tot["GRBM_GUI_ACTIVE"] / 8 * 1024)
            line += f"  mfma_util~{util:.3f}"
        print(line)
