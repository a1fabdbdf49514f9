#!/bin/bash
# Kernel trace of one launch sequence on the moving-objects scene (run on the GPU box from the repository root):
#   bash profiles/hard_trace.sh <tag>   ->  gpurun_out/<tag>_hard_scene_kernel_stats.csv
set -e
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_hard
export VO_NO_DIAG=1
timeout -s KILL 250 rocprofv3 --kernel-trace --stats -d /tmp/prof_hard -- python3 $R/profiles/diag/stamp_ransac_hard.py 30 32 > $OUT/${TAG}_hard_scene_hist.txt 2>&1
python3 $R/profiles/kstats.py $(find /tmp/prof_hard -name "*.db" | head -n 1) > $OUT/${TAG}_hard_scene_kernel_stats.csv
echo collected
