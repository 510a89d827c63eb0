#!/usr/bin/env python3
"""Where a (graph-replayed, multi-stream) step's wall time goes: busy union vs idle gaps, from a rocprofv3 kernel trace of bench.py.

    python tools/prof_gaps.py gpurun_out/prof_<tag> [--last-steps N]

Steps are delimited by the Adam kernel (one per step).  Per step: wall span, the union of all kernel intervals (GPU busy with at least
one kernel), the sum of kernel durations (> union where streams overlap), the idle time, and the idle time charged to the kernel that
STARTS after each gap (top of the list = where launch latency / dependencies hurt)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
n_last = int(sys.argv[sys.argv.index("--last-steps") + 1]) if "--last-steps" in sys.argv else 4
db = sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
ends = [r[0] for r in c.execute('select "end" from kernels where name like \'%adam_flat%\' order by "end"').fetchall()]
assert len(ends) > n_last, f"only {len(ends)} Adam launches in the trace"
rows = c.execute(f'select name, start, "end" from kernels where start > {ends[-n_last - 1]} and "end" <= {ends[-1]} order by start').fetchall()


def short(name):
    s = name.split("(")[0].replace("void ", "").replace("cd::", "")
    return s[:s.index("<")] if "<" in s else s[-50:]


span = (ends[-1] - ends[-n_last - 1]) / 1e6
busy = idle = 0.0
cur_end = ends[-n_last - 1]
gap_after, gap_n, hist = defaultdict(float), defaultdict(int), defaultdict(int)
for name, s, e in rows:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        idle += g
        gap_after[short(name)] += g
        gap_n[short(name)] += 1
        hist[min(int(g // 2) * 2, 20)] += 1
    busy += max(0, e - max(s, cur_end)) / 1e3
    cur_end = max(cur_end, e)
total = sum(e - s for _, s, e in rows) / 1e3
print(f"{n_last} steps: wall {span / n_last:.3f} ms/step, busy union {busy / 1e3 / n_last:.3f}, sum of kernel durations {total / 1e3 / n_last:.3f}, "
      f"idle {idle / 1e3 / n_last:.3f} ms/step over {sum(gap_n.values()) / n_last:.0f} gaps/step, {len(rows) / n_last:.0f} launches/step")
print("gap histogram (us bucket: count/step): " + "  ".join(f"{k}{'+' if k == 20 else ''}:{v / n_last:.0f}" for k, v in sorted(hist.items())))
print("idle time charged to the kernel that starts after the gap (us/step, gaps/step):")
for k, v in sorted(gap_after.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {k:44s} {v / n_last:8.1f} us  {gap_n[k] / n_last:6.1f}")
