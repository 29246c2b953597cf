#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
date > gpurun_out/r05_u.log
timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>>gpurun_out/r05_u.log | tail -1 > gpurun_out/r05_u_config5.json
echo "bench rc=$? $(date)" >> gpurun_out/r05_u.log
( cd /tmp && rm -rf /tmp/prof_t5 && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
echo "rocprof rc=$? $(date)" >> gpurun_out/r05_u.log
tail -3 /tmp/prof_t5.log >> gpurun_out/r05_u.log
timeout 120 python tools/step_timeline.py $(find /tmp/prof_t5 -name "*results.db" | head -1) k_af_a 5 10 > gpurun_out/r05_config5_timeline.txt 2>&1
echo "timeline rc=$? $(date)" >> gpurun_out/r05_u.log
