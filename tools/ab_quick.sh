#!/bin/bash
# quick A/B of the default workload: prints fps, ms per step and the per-stage times (run on the GPU box)
#   bash tools/ab_quick.sh [extra bench.py arguments]
python bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), d['stage_ms_per_launch'])"
