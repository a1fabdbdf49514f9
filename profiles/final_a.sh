#!/bin/bash
# First half of a round's profiles (run on the GPU box from the repository root, after `make -C profiles/diag` and
# `make -C profiles/diag cuts` here): every rocprofv3 pass.  profiles/traffic.py then turns the counter files into
# rNN_traffic.json (on any machine), and profiles/final_b.sh takes the bench lines that quote it.
#   bash profiles/final_a.sh <tag>
set -e
TAG=${1:-r03}
R=$(pwd)
bash $R/profiles/collect.sh $TAG
cd $R
bash $R/profiles/knn_counters.sh $TAG > /dev/null
cd $R
bash $R/profiles/phase_counts.sh $TAG
cd $R
bash $R/profiles/hard_trace.sh $TAG
cd $R
bash $R/profiles/latency_trace.sh ${TAG}_pageable fast
cd $R
bash $R/profiles/latency_trace.sh ${TAG}_pinned fast pinned
