#!/usr/bin/env python3
"""Summarise a tools/prof_loss.sh output directory (rocprofv3 rocpd SQLite databases): per-kernel
average duration from the kernel trace and the per-dispatch mean of every PMC counter.

    python tools/prof_summary.py gpurun_out/prof_<tag> [kernel-name-substring]
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
focus = sys.argv[2] if len(sys.argv) > 2 else ""


def short(n):
    return n.split("(")[0].replace("void ", "")[-64:]


for db in sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    print("== kernel trace", os.path.relpath(db, root))
    rows = c.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration), vgpr_count, lds_size "
                     "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
    for name, gx, gy, gz, n, avg, mn, vg, lds in rows:
        print(f"  {short(name):64s} grid=({gx},{gy},{gz}) n={n:4d} avg_us={avg / 1e3:10.2f} min_us={mn / 1e3:10.2f} vgpr={vg} lds={lds}")
for db in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    d = defaultdict(lambda: defaultdict(list))
    try:
        rows = c.execute("select name, counter_name, counter_value, dispatch_id from pmc_events").fetchall()
    except sqlite3.Error as e:
        print("== pmc", db, "unreadable:", e)
        continue
    per_dispatch = defaultdict(float)
    kname = {}
    for name, cn, cv, did in rows:
        per_dispatch[(did, cn)] += cv  # counters come per instance (XCD/SE): sum them per dispatch
        kname[did] = short(name)
    for (did, cn), v in per_dispatch.items():
        d[kname[did]][cn].append(v)
    print("== pmc", os.path.relpath(db, root))
    for k, cs in d.items():
        if focus and focus not in k:
            continue
        print("  " + k)
        for cn, v in sorted(cs.items()):
            print(f"      {cn:28s} mean/dispatch={sum(v) / len(v):18.1f}  n={len(v)}")
