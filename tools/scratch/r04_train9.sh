#!/bin/bash
# hybrid training: heavy rows of the remainder CSR -- parity tests, the exophormer training lines, kernel statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
set -x
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 > gpurun_out/r04_bench_config_5_exophormer_d539.json 2> gpurun_out/r04_bench_config_5_exophormer_d539.err
timeout 600 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision bf16 > gpurun_out/r04_bench_config_5_exophormer_d539_bf16mma.json 2> gpurun_out/r04_bench_config_5_exophormer_d539_bf16mma.err
for f in config_5_exophormer_d539 config_5_exophormer_d539_bf16mma; do python -c "import json; d=json.loads(open('gpurun_out/r04_bench_$f.json').read().strip().split('\n')[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'])"; done
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt 2>&1
head -30 gpurun_out/r04_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt | cut -c1-150
} > gpurun_out/r04_train9.log 2>&1
