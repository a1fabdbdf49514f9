#!/bin/bash
# FETCH_SIZE of the tile kernels under the two XCD tile orders (column-major: the tiles of a tile column share an
# XCD; row-major: the tiles of a tile row do).  Run on the GPU box from the repository root.
set -e
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 3 --warmup 1"
for order in cols rows; do
  rm -rf /tmp/prof_to_$order
  export VO_TILE_ORDER=$order
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_to_$order -- python3 $R/bench.py $ARGS > /tmp/to_$order.log 2>&1
  echo "== VO_TILE_ORDER=$order"
  python3 $R/profiles/pmcstats.py $(find /tmp/prof_to_$order -name "*.db" | head -n 1) | grep -E "k_frontend|k_resize"
done > $OUT/${TAG}_tile_order_fetch.txt
cat $OUT/${TAG}_tile_order_fetch.txt
