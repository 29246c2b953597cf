#!/bin/bash
# bf16-mode fixture test; kernel statistics of the exophormer training step (bf16 operands); the from-pixels line with bf16 maps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
set -x
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -s -k "bf16_mode_vs_reference_fixture or real_ddp_wrapper_two_ranks" 2>&1 | grep -E "worst|passed|failed|Error|assert" | head -20
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt 2>&1
head -40 gpurun_out/r04_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt | cut -c1-150
timeout 900 python bench.py --config 5 --pixels --precision bf16 > gpurun_out/r04_bench_config_5_pixels_bf16.json 2> gpurun_out/r04_bench_config_5_pixels_bf16.err
python -c "import json; d=json.loads(open('gpurun_out/r04_bench_config_5_pixels_bf16.json').read().strip().split('\n')[-1]); print('pixels bf16', round(d['value']), round(d['ms_per_step'],3), d.get('phases_ms'))"
} > gpurun_out/r04_train8.log 2>&1
