#!/bin/bash
# fused small-group attention (k_attn_small_fwd / bwd): parity tests of the training path + config 5 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
set -x
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_samplers.py -x -q -m gpu -k "without_the_mfma_hoist" 2>&1 | tail -5
for rep in 1 2; do
for cfg in "DA_X=1" "DA_X=2"; do
  env $cfg timeout 300 python bench.py --config 5 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16 $cfg', round(d['value']), round(d['ms_per_step'],3), d.get('phases_ms'))"
done
done
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --precision bf16 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
head -40 gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt | cut -c1-150
} > gpurun_out/r04_train6.log 2>&1
