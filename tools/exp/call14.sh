mkdir -p gpurun_out/r3
for g in 1 0; do
  CD_AMD_MIDAS_GRAPH=$g timeout 400 python bench.py --model midas2 --height 384 --width 384 --batch-size 8 --steps 8 --warmup 4 --no-loss-microbench --frames 20 > gpurun_out/r3/bench_midas_g$g.json 2> gpurun_out/r3/bench_midas_g$g.err
  tail -4 gpurun_out/r3/bench_midas_g$g.err
done
python - <<'PY'
import json
for g in (1,0):
    try:
        d=json.loads(open(f'gpurun_out/r3/bench_midas_g{g}.json').read().strip().splitlines()[-1]); print(g, d['value'], d['ms_per_step'], d['config']['hip_graph'], d['config']['last_loss'])
    except Exception as e: print(g, 'ERR', e)
PY
