#!/usr/bin/env python3
"""Top kernels by total time from a rocprofv3 kernel trace (rocpd sqlite) of bench.py."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
db = sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
print(f"kernel time total {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} launches; trace span {(t1 - t0) / 1e6:.2f} ms")
for name, n, tot, avg in rows[:28]:
    short = name.split("(")[0].replace("void ", "")[-78:]
    print(f"  {short:78s} n={n:6d} total_ms={tot / 1e6:9.3f} ({100 * tot / total:5.1f}%) avg_us={avg / 1e3:9.2f}")
