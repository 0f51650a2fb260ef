mkdir -p gpurun_out/r5v
for g in 0 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-k1-512 --no-ceiling --global-iters $g --steps 20 --warmup 4 > gpurun_out/r5v/bench_g$g.json 2> gpurun_out/r5v/bench_g$g.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5v/bench_g$g.json').read().strip().splitlines()[-1])
for k in ('frame','frame_512'):
    v=d[k]; print('global_iters $g', k, round(v['ms_per_frame'],3), {a:round(b,3) for a,b in v['stage_ms_with_syncs'].items()}, v['samples'])
PY
done
