#!/bin/bash
# Effective shader clock of the kernels of the default bench workload: GRBM_GUI_ACTIVE (summed over the 8 XCDs)
# / 8 / kernel duration.  Run on the GPU box from the repository root:  bash profiles/clock_once.sh <tag>
set -e
TAG=${1:-r02}
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_clk /tmp/prof_clk_kt
ARGS="--hard-scene 0 --flat-scene 0 --host-fed 0 --facade-latency 0 --cpu-frames 0 --cpu-workers 0 --steps 3 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE -d /tmp/prof_clk -- python3 $R/bench.py $ARGS > /tmp/clk.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_clk_kt -- python3 $R/bench.py $ARGS > /tmp/clk_kt.log 2>&1
python3 $R/profiles/pmcstats.py $(find /tmp/prof_clk -name "*.db" | head -n 1) > /tmp/clk_pmc.txt
python3 $R/profiles/kstats.py $(find /tmp/prof_clk_kt -name "*.db" | head -n 1) > /tmp/clk_kt.csv
python3 - <<PY > $R/gpurun_out/${TAG}_effective_clock.txt
import csv
act = {}
for line in open("/tmp/clk_pmc.txt"):
    p = line.split()
    if len(p) >= 3 and "GRBM_GUI_ACTIVE=" in p[-1]:
        act[p[0][:48]] = float(p[-1].split("=")[1])
dur = {}
for row in csv.DictReader(open("/tmp/clk_kt.csv")):
    dur[row["kernel"][:48]] = float(row["avg_us"])
print("kernel, avg_us (kernel trace), GRBM_GUI_ACTIVE per dispatch (sum of 8 XCDs), effective clock GHz")
for k, a in act.items():
    if k in dur and dur[k] > 20:
        print(f"{k}, {dur[k]:.1f}, {a:.4g}, {a / 8 / dur[k] / 1e3:.3f}")
PY
cat $R/gpurun_out/${TAG}_effective_clock.txt
