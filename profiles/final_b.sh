#!/bin/bash
# Second half of a round's profiles (run on the GPU box from the repository root): the three bench lines with all
# legs on, the secondary configurations and the single-frame switches.
#   bash profiles/final_b.sh <tag>
set -e
TAG=${1:-r03}
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench_orb.json 2> gpurun_out/${TAG}_bench_orb.err
timeout -k 10 300 python3 bench.py --detector fast > gpurun_out/${TAG}_bench_fast.json 2> gpurun_out/${TAG}_bench_fast.err
timeout -k 10 200 python3 bench.py --matcher popcount --cpu-frames 0 --cpu-workers 0 > gpurun_out/${TAG}_bench_orb_popcount.json 2> gpurun_out/${TAG}_bench_popcount.err
cat gpurun_out/${TAG}_bench_orb.json gpurun_out/${TAG}_bench_fast.json gpurun_out/${TAG}_bench_orb_popcount.json > gpurun_out/${TAG}_bench_lines.jsonl
timeout -k 10 300 python3 tools/bench_configs.py > gpurun_out/${TAG}_secondary_configs.jsonl 2> gpurun_out/${TAG}_configs.err
timeout -k 10 400 python3 -u tools/lat_stage_ab.py > gpurun_out/${TAG}_latency_staging_ab.txt 2>&1
timeout -k 10 300 python3 -u tools/lat_depth_ab.py > gpurun_out/${TAG}_depth_in_place_ab.txt 2>&1
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_lines.jsonl"):
    d = json.loads(l)
    print(d["config"]["detector"], d["config"]["matcher"], round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d.get("cpu_baseline", {}).get("value"))
PY
