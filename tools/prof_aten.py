#!/usr/bin/env python3
"""Every kernel of the last N steps of a bench.py trace that is NOT one of this package's (namespace cd::): full names, grids, counts per
step -- what of the framework still runs inside the step.   python tools/prof_aten.py gpurun_out/prof_<tag> [N]"""
import glob
import os
import sqlite3
import sys

root, n_last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
db = sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
ends = [r[0] for r in c.execute('select "end" from kernels where name like \'%adam_flat%\' order by "end"').fetchall()]
where = f' where start > {ends[-n_last - 1]} and "end" <= {ends[-1]}'
rows = c.execute(f"select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration) from kernels{where} group by name, grid_x, grid_y, grid_z "
                 "order by count(*) desc").fetchall()
tot_n = tot_t = 0
for name, gx, gy, gz, wx, n, t in rows:
    if "cd::" in name:
        continue
    tot_n += n
    tot_t += t
    print(f"{n / n_last:7.2f}/step {t / 1e3 / n_last:8.2f} us/step  blocks=({gx // max(wx, 1)},{gy},{gz}) wg={wx}  {name[:400]}")
print(f"total: {tot_n / n_last:.1f} launches/step, {tot_t / 1e3 / n_last:.1f} us/step of non-cd:: kernels")
