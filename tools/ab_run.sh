#!/bin/bash
# A/B builds of the library on one GPU box: tools/ab_run.sh <lib_a> <lib_b> [<lib_c> ...] [-- bench flags]
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in 1 2; do
  for l in "${libs[@]}"; do
    timeout -k 10 100 python tools/ab_bench.py "$l" --cpu-workers 0 --cpu-frames 0 "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -n 5 /tmp/ab.err; exit 1; }
    python - "$l" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 3), {k: v for k, v in d["stage_ms_per_launch"].items() if v > 0.05})
PY
  done
done
