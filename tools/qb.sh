timeout -k 10 200 python bench.py --host-fed 0 --cpu-frames 0 --cpu-workers 0 --hard-scene 0 --flat-scene 0 --facade-latency 0 --steps 30 > gpurun_out/qb.json 2>gpurun_out/qb.err || { tail -5 gpurun_out/qb.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/qb.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], 'select', d['stage_ms_per_launch']['select_topn'], 'brief', d['stage_ms_per_launch']['brief'])
PY
