#!/usr/bin/env python3
"""Top kernels by total time from a rocprofv3 kernel trace (rocpd sqlite) of bench.py.

    python tools/prof_step_summary.py gpurun_out/prof_<tag> [steps] [--by-grid]

With --by-grid the rows are (kernel, grid, workgroup) so that every layer shape shows up separately.
With --last-steps N only the last N optimisation steps are counted (delimited by the Adam kernel, one per step),
which drops the warm-up steps and the convolution autotuner's timing launches.
"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
by_grid = "--by-grid" in sys.argv
nums = [a for i, a in enumerate(sys.argv[2:], 2) if a.isdigit() and sys.argv[i - 1] != "--last-steps"]
steps = int(nums[0]) if nums else None
db = sorted(glob.glob(os.path.join(root, "trace", "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
where = ""
if "--last-steps" in sys.argv:
    n_last = int(sys.argv[sys.argv.index("--last-steps") + 1])
    ends = [r[0] for r in c.execute('select "end" from kernels where name like \'%adam_flat%\' order by "end"').fetchall()]
    assert len(ends) > n_last, f"only {len(ends)} Adam launches in the trace"
    where = f' where start > {ends[-n_last - 1]} and "end" <= {ends[-1]}'
    steps = n_last
    nums = []
grp = "name, grid_x, grid_y, grid_z, workgroup_x" if by_grid else "name"
sel = "name, count(*), sum(duration), avg(duration)" + (", grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count" if by_grid else "")
rows = c.execute(f"select {sel} from kernels{where} group by {grp} order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
t0, t1 = c.execute(f"select min(start), max(end) from kernels{where}").fetchone()
print(f"kernel time total {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} launches; trace span {(t1 - t0) / 1e6:.2f} ms"
      + (f"; per step (/{steps}): {total / 1e6 / steps:.2f} ms" if steps else ""))
for r in rows[:(90 if by_grid else 200)]:
    name, n, tot, avg = r[:4]
    short = name.split("(")[0].replace("void ", "").replace("cd::", "")
    if "<" in name.split("(")[0]:
        short = name[:name.index("(")].replace("void ", "").replace("cd::", "")
    short = short[-60:]
    extra = ""
    if by_grid:
        gx, gy, gz, wx, lds, vg = r[4:]
        extra = f" blocks=({gx // max(wx, 1)},{gy},{gz}) wg={wx} lds={lds} vgpr={vg}"
    per = f" per_step_ms={tot / 1e6 / steps:7.3f}" if steps else ""
    print(f"  {short:60s} n={n:6d} total_ms={tot / 1e6:9.3f} ({100 * tot / total:5.1f}%) avg_us={avg / 1e3:9.2f}{per}{extra}")
