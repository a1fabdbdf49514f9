#!/bin/bash
# One kernel-trace pass of the default workload (run on the GPU box from the repository root):
#   bash profiles/kt_once.sh <tag> [extra bench.py arguments]  ->  gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=${1:-kt}
shift || true
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt_once
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_once -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 10 --warmup 2 "$@" > /tmp/kt_once.log 2>&1
python3 $R/profiles/kstats.py $(find /tmp/prof_kt_once -name "*.db" | head -n 1) > $OUT/${TAG}_kernel_stats.csv
echo collected
