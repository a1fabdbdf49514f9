#!/bin/bash
# Collect the rocprofv3 summaries kept in profiles/ (run on the GPU box from the repository root):
#   bash profiles/collect.sh <tag>      e.g. r01  ->  gpurun_out/<tag>_*.{csv,txt}
# One rocprofv3 run per pass (kernel trace; FETCH_SIZE; WRITE_SIZE; SQ counters), the program itself after `--`.
set -e
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_orb -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --steps 10 --warmup 2 > /tmp/kt_orb.log 2>&1
python3 $R/profiles/kstats.py $(find /tmp/prof_kt_orb -name "*.db" | head -n 1) > $OUT/${TAG}_bench_orb_b512_kernel_stats.csv
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_fast -- python3 $R/bench.py --detector fast --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --steps 10 --warmup 2 > /tmp/kt_fast.log 2>&1
python3 $R/profiles/kstats.py $(find /tmp/prof_kt_fast -name "*.db" | head -n 1) > $OUT/${TAG}_bench_fast_b512_kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --steps 3 --warmup 1 > /tmp/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --steps 3 --warmup 1 > /tmp/write.log 2>&1
{ python3 $R/profiles/pmcstats.py $(find /tmp/prof_fetch -name "*.db" | head -n 1); python3 $R/profiles/pmcstats.py $(find /tmp/prof_write -name "*.db" | head -n 1); } > $OUT/${TAG}_bench_orb_b512_pmc_hbm.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES -d /tmp/prof_sq -- python3 $R/bench.py --hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --steps 3 --warmup 1 > /tmp/sq.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_sq -name "*.db" | head -n 1) > $OUT/${TAG}_bench_orb_b512_pmc_sq.txt
echo collected
