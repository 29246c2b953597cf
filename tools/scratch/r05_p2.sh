#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for bf in 1 0; do
( cd /tmp && rm -rf /tmp/prof_p && DA_PCD_TWO_STREAMS=0 DA_PCD_KNN_BF3=$bf timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o s -- python $GRAFT_REPO_ROOT/bench.py --mode encode --config 4 --no-cpu-baseline > /tmp/prof_p.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_p -name "*results.db" | head -1) 2>&1 | head -14 > gpurun_out/r05_p2_stats_bf3_$bf.txt
done
