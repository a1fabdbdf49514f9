#!/bin/bash
# Per-pyramid-level duration and instruction counts of the tile kernel: bash profiles/levels.sh  (on the GPU box)
set -e
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lv_*
timeout -s KILL 200 rocprofv3 --kernel-trace -d /tmp/prof_lv_kt -- python3 $R/bench.py --cpu-frames 0 --cpu-workers 0 --steps 5 --warmup 2 > /tmp/lv_kt.log 2>&1
python3 $R/profiles/levels.py $(find /tmp/prof_lv_kt -name "*.db" | head -n 1) k_frontend > $R/gpurun_out/levels_time.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d /tmp/prof_lv_sq -- python3 $R/bench.py --cpu-frames 0 --cpu-workers 0 --steps 2 --warmup 1 > /tmp/lv_sq.log 2>&1
python3 $R/profiles/levels.py $(find /tmp/prof_lv_sq -name "*.db" | head -n 1) k_frontend > $R/gpurun_out/levels_sq.txt
cat $R/gpurun_out/levels_time.txt $R/gpurun_out/levels_sq.txt
