#!/bin/bash
# Issue / stall counters of the matcher kernel: bash profiles/knn_counters.sh  (on the GPU box)
set -e
TAG=${1:-r02}
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_knn_*
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/prof_knn_a -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 2 --warmup 1 > /tmp/knn_a.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_knn_a -name "*.db" | head -n 1) k_knn2 > $R/gpurun_out/${TAG}_matcher_counters.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE -d /tmp/prof_knn_b -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 2 --warmup 1 > /tmp/knn_b.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_knn_b -name "*.db" | head -n 1) k_knn2 >> $R/gpurun_out/${TAG}_matcher_counters.txt
cat $R/gpurun_out/${TAG}_matcher_counters.txt
