#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_px -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --pixels --precision bf16 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_px.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_px -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_config5_pixels_bf16.txt 2>&1
head -45 gpurun_out/r04_rocprof_kernel_stats_config5_pixels_bf16.txt | cut -c1-170
