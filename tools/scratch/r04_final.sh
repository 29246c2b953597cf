#!/bin/bash
# end of round: full GPU suite, smoke, and the training lines re-collected with the last kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --config 5 > $O/r04_bench_config_5.json 2> $O/r04_bench_config_5.err
timeout 600 python bench.py --config 5 --precision fp32 > $O/r04_bench_config_5_fp32.json 2> $O/r04_bench_config_5_fp32.err
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_t5 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $O/r04_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
for f in config_5 config_5_fp32; do python -c "import json; d=json.loads(open('$O/r04_bench_$f.json').read().strip().split('\n')[-1]); print('$f', round(d['value']), round(d['ms_per_step'],4), d['roofline']['frac'])"; done
head -12 $O/r04_rocprof_kernel_stats_config5_bf16mma.txt | cut -c1-140
} > $O/r04_final.log 2>&1
