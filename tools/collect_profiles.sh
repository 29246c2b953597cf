#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel statistics and PMC traffic of the bench configurations, summarised into
# gpurun_out/<round>_*.txt (copied to profiles/<round>/ afterwards; ROUND env, default r03).  Counters are collected in their own passes with
# --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
#   usage: bash tools/collect_profiles.sh <tag> <bench args...>      e.g.  headline --config 3p
set -u
ROUND=${ROUND:-r06}
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $W
cd /tmp
# DA_TWO_BRANCH=0: every launch at the full Batch size, so that per-launch averages (durations, PMC KiB) describe ONE kernel shape --
# the shape bench.py's `roofline` object is about (its per-class times come from an eager one-branch pass); the default two-branch
# loop launches every kernel twice at half size and, under --kernel-trace, runs the branches serially (tools/graph_idle_probe.py).
# profiles/r03/r03_rocprof_kernel_stats_headline_two_branch.txt is the same collection with the default loop.
export DA_TWO_BRANCH=${DA_TWO_BRANCH:-0}
BENCH="python $REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --replays 0 $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $W/stats -o s -- $BENCH > $W/stats.log 2>&1
DB=$(find $W/stats -name "*results.db" | head -1)
python $REPO/profiles/rocpd_stats.py $DB > $OUT/${ROUND}_rocprof_kernel_stats_$TAG.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $W/$C -o p -- $BENCH > $W/$C.log 2>&1
  DB=$(find $W/$C -name "*results.db" | head -1)
  python $REPO/profiles/rocpd_pmc.py $DB k_ >> $OUT/${ROUND}_pmc_traffic_$TAG.txt 2>&1
done
tail -2 $W/stats.log | head -1 | cut -c1-300
echo "== $TAG done"
