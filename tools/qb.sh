#!/bin/bash
# Quick look at one build on the GPU box: the headline leg only, 30 steps, every stage time.
#   bash tools/qb.sh        (VO_LIB_PATH=<other libvo_hip.so> for a diagnostic build)
timeout -k 10 200 python bench.py --host-fed 0 --cpu-frames 0 --cpu-workers 0 --hard-scene 0 --flat-scene 0 --facade-latency 0 --steps 30 > gpurun_out/qb.json 2>gpurun_out/qb.err || { tail -5 gpurun_out/qb.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/qb.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'], 3), d['stage_ms_per_launch'])
PY
