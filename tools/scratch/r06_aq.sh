#!/bin/bash
# GPU box: rocprofv3 kernel statistics of BASELINE configuration 4 on HEAD (k_attn_tiny in the last layer)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_c4 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --replays 0 > /tmp/prof_c4.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_c4 -name "*results.db" | head -1) > gpurun_out/r06_rocprof_kernel_stats_config4.txt 2>&1
head -16 gpurun_out/r06_rocprof_kernel_stats_config4.txt | cut -c1-150
