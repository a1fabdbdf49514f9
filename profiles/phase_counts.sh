#!/bin/bash
# Instruction counters of the tile kernels by phase (run on the GPU box from the repository root, after
# `make -C profiles/diag cuts` here): the kernels are built to stop after their first / second / third barrier
# (VO_FE_CUT), and each variant runs 256 HD720 frames of the full detector under one rocprofv3 --pmc pass.
#   bash profiles/phase_counts.sh <tag>  ->  gpurun_out/<tag>_tile_kernel_phase_counts.txt
set -e
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out/${TAG}_tile_kernel_phase_counts.txt
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
for v in cut1 cut2 cut3 full; do
    lib=$R/profiles/diag/libvo_hip_$v.so
    [ $v = full ] && lib=$R/visual_odometry_amd/libvo_hip.so
    rm -rf /tmp/prof_pc_$v
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d /tmp/prof_pc_$v -- python3 $R/profiles/diag/count_frontend.py $lib 8 > /tmp/pc_$v.log 2>&1
    echo $v >> $OUT
    python3 $R/profiles/pmcstats.py $(find /tmp/prof_pc_$v -name "*.db" | head -n 1) | grep k_frontend >> $OUT
done
echo collected
