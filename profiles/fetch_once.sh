#!/bin/bash
# One FETCH_SIZE pass of the default bench workload (run on the GPU box from the repository root):
#   bash profiles/fetch_once.sh <tag>  ->  gpurun_out/<tag>_fetch.txt
set -e
TAG=${1:-x}
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f1 -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 3 --warmup 1 > /tmp/f1.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_f1 -name "*.db" | head -n 1) > $R/gpurun_out/${TAG}_fetch.txt
head -n 8 $R/gpurun_out/${TAG}_fetch.txt
