#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the level-0 tile kernel on BGRA and on gray input (tools/calib_fetch.py), run on the GPU
# box from the repository root:  bash profiles/calib.sh <tag>  ->  gpurun_out/<tag>_fetch_calibration.txt
set -e
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal_*
for C in 4 1; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $CNT -d /tmp/cal_${C}_$CNT -- python3 $R/tools/calib_fetch.py --channels $C > /tmp/cal_${C}_$CNT.log 2>&1
  done
done
{
  for C in 4 1; do
    echo "# channels=$C"
    for CNT in FETCH_SIZE WRITE_SIZE; do
      python3 $R/profiles/pmcstats.py $(find /tmp/cal_${C}_$CNT -name "*.db" | head -n 1) k_frontend
    done
  done
} > $OUT/${TAG}_fetch_calibration.txt
cat $OUT/${TAG}_fetch_calibration.txt
