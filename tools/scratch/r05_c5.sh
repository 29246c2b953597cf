#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --config 5 > $O/r05_bench_config_5.json 2> $O/r05_bench_config_5.err
timeout 900 python bench.py --config 5 --precision fp32 > $O/r05_bench_config_5_fp32.json 2> $O/r05_bench_config_5_fp32.err
timeout 900 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 > $O/r05_bench_config_5_exophormer_d539.json 2>/dev/null
( cd /tmp && rm -rf /tmp/prof_t5 && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $O/r05_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
timeout 120 python tools/step_timeline.py $(find /tmp/prof_t5 -name "*results.db" | head -1) k_af_a 5 10 > $O/r05_config5_step_timeline_final.txt 2>&1
