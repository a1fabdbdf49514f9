#!/bin/bash
# Instruction-cache counters of the kernels of the default bench workload (run on the GPU box from the repository
# root):  bash profiles/icache_once.sh <tag>  ->  gpurun_out/<tag>_icache.txt
set -e
TAG=${1:-r03}
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ic
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES -d /tmp/prof_ic -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 3 --warmup 1 > /tmp/ic.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_ic -name "*.db" | head -n 1) > $R/gpurun_out/${TAG}_icache.txt
echo collected
