#!/bin/bash
# One kernel-trace pass of a secondary configuration (run on the GPU box from the repository root):
#   bash profiles/kt_config.sh <tag> <stereo1080|kitti|estress>  ->  gpurun_out/<tag>_<config>_kernel_stats.csv
set -e
TAG=${1:-kt}
CFG=${2:-stereo1080}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt_cfg
timeout -s KILL 250 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_cfg -- python3 $R/tools/bench_configs.py --configs $CFG --steps 6 --warmup 2 > /tmp/kt_cfg.log 2>&1
python3 $R/profiles/kstats.py $(find /tmp/prof_kt_cfg -name "*.db" | head -n 1) > $OUT/${TAG}_${CFG}_kernel_stats.csv
echo collected
