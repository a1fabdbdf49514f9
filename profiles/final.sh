#!/bin/bash
# Everything profiles/ keeps for a round, from one box and session (run on the GPU box from the repository root):
#   bash profiles/final.sh <tag>  ->  gpurun_out/<tag>_*  (rocprofv3 summaries, the three bench lines, secondary configs)
set -e
TAG=${1:-r01}
R=$(pwd)
bash $R/profiles/collect.sh $TAG
cd $R
bash $R/profiles/knn_counters.sh $TAG > /dev/null
cd $R
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench_orb.json 2> gpurun_out/${TAG}_bench_orb.err
timeout -k 10 300 python3 bench.py --detector fast > gpurun_out/${TAG}_bench_fast.json 2> gpurun_out/${TAG}_bench_fast.err
timeout -k 10 200 python3 bench.py --matcher popcount --cpu-frames 0 --cpu-workers 0 > gpurun_out/${TAG}_bench_orb_popcount.json 2> gpurun_out/${TAG}_bench_popcount.err
cat gpurun_out/${TAG}_bench_orb.json gpurun_out/${TAG}_bench_fast.json gpurun_out/${TAG}_bench_orb_popcount.json > gpurun_out/${TAG}_bench_lines.jsonl
timeout -k 10 300 python3 tools/bench_configs.py > gpurun_out/${TAG}_secondary_configs.jsonl 2> gpurun_out/${TAG}_configs.err
python3 - <<PY
import json
for l in open("gpurun_out/${TAG}_bench_lines.jsonl"):
    d = json.loads(l)
    print(d["config"]["detector"], d["config"]["matcher"], round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d.get("cpu_baseline", {}).get("value"))
PY
