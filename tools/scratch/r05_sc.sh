#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_sc && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sc -o s -- python $GRAFT_REPO_ROOT/bench.py --config scripted --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_sc.log 2>&1 )
timeout 120 python tools/step_timeline.py $(find /tmp/prof_sc -name "*results.db" | head -1) k_af_a 3 6 > $O/r05_scripted_train_timeline.txt 2>&1
tail -3 /tmp/prof_sc.log >> $O/r05_scripted_train_timeline.txt
