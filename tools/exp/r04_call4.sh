#!/bin/bash
set -u
export CD_AMD_CONV_TUNE_CACHE=$PWD/gpurun_out/conv_tune.json
python bench.py --no-cpu-baseline --no-config5 --no-loss-microbench --steps 2 --warmup 3 > /dev/null 2>&1
CD_AMD_WGRAD_BATCH=0 CD_AMD_ENGINE_STREAMS=none bash tools/prof_step.sh bg --backend hip --steps 4 --warmup 3 --graph 0 --no-cpu-baseline --no-config5 --no-loss-microbench > /dev/null 2>&1
python tools/prof_step_summary.py gpurun_out/prof_bg --last-steps 4 --by-grid > gpurun_out/bygrid.txt 2>&1
python - <<'PY' > gpurun_out/bygrid_all.txt
import glob, sqlite3
db = sorted(glob.glob("gpurun_out/prof_bg/trace/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
ends = [r[0] for r in c.execute('select "end" from kernels where name like \'%adam_flat%\' order by "end"').fetchall()]
where = f' where start > {ends[-5]} and "end" <= {ends[-1]}'
rows = c.execute(f"select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration) from kernels{where} group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc").fetchall()
for name, gx, gy, gz, wx, n, tot, avg in rows:
    short = name[:name.index("(")].replace("void ", "").replace("cd::", "") if "(" in name else name
    print(f"{short[-56:]:56s} blocks=({gx // max(wx,1)},{gy},{gz}) n={n} per_step_us={tot/4e3:9.1f} avg_us={avg/1e3:8.2f}")
PY
rm -rf gpurun_out/prof_bg/trace
wc -l gpurun_out/bygrid_all.txt
