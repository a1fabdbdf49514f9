#!/bin/bash
# Kernel + copy timeline of one processFrame call through the C++ facade (run on the GPU box from the repository
# root):  bash profiles/latency_trace.sh <tag> [fast|orb] [pinned]  ->  gpurun_out/<tag>_latency_timeline.txt
set -e
TAG=${1:-lat}
DET=${2:-fast}
PIN=${3:-}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np
from visual_odometry_amd import synth
cam = synth.CONFIG_CAMERAS["hd720"]
f, d, _ = synth.make_pingpong_batch(cam, 1, 8, 2000, channels=4, relief=2.0)
with open("/tmp/lat_dump.bin", "wb") as fh:
    fh.write(np.ascontiguousarray(f).tobytes())
    fh.write(np.ascontiguousarray(d, np.float32).tobytes())
open("/tmp/lat_args.txt", "w").write(" ".join([str(cam.width), str(cam.height), "4", "8", repr(cam.fx), repr(cam.fy), repr(cam.cx), repr(cam.cy)]))
PY
ARGS=$(cat /tmp/lat_args.txt)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lat
timeout -s KILL 120 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_lat -- $R/tools/facade_bench /tmp/lat_dump.bin $ARGS 40 $DET $PIN > /tmp/lat.log 2>&1
python3 $R/profiles/latency_timeline.py $(find /tmp/prof_lat -name "*.db" | head -n 1) > $OUT/${TAG}_latency_timeline.txt
tail -n 2 /tmp/lat.log >> $OUT/${TAG}_latency_timeline.txt
echo collected
