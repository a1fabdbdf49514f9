#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/fetch_probe (three kernels that each read a 512 MiB buffer exactly once with one
# load shape): run on the GPU box from the repository root ->  gpurun_out/<tag>_fetch_probe.txt
set -e
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_probe_*
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_probe_f -- $R/tools/fetch_probe 512 > /tmp/probe_f.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_probe_w -- $R/tools/fetch_probe 512 > /tmp/probe_w.log 2>&1
{
  grep "bytes read" /tmp/probe_f.log
  python3 $R/profiles/pmcstats.py $(find /tmp/prof_probe_f -name "*.db" | head -n 1)
  python3 $R/profiles/pmcstats.py $(find /tmp/prof_probe_w -name "*.db" | head -n 1)
} > $OUT/${TAG}_fetch_probe.txt
cat $OUT/${TAG}_fetch_probe.txt
