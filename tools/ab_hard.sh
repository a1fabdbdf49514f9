#!/bin/bash
# A/B builds of the library on one GPU box, headline and moving-objects scene: tools/ab_hard.sh <lib_a> <lib_b> ...
for i in 1 2; do
  for l in "$@"; do
    timeout -k 10 200 python tools/ab_bench.py "$l" --cpu-workers 0 --cpu-frames 0 --host-fed 0 --facade-latency 0 > /tmp/ab.json 2>/tmp/ab.err || { tail -n 5 /tmp/ab.err; exit 1; }
    python - "$l" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], "headline", round(d["value"]), d["stage_ms_per_launch"]["pnp_ransac"], "hard", round(d["hard_scene"]["value"]), d["hard_scene"]["stage_ms_per_launch"]["pnp_ransac"], d["hard_scene"]["stage_ms_per_launch"]["f_ransac"], "flat", round(d["flat_scene"]["value"]), d["flat_scene"]["stage_ms_per_launch"]["pnp_ransac"])
PY
  done
done
