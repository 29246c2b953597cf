#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_rccl.py tests/test_gpu_encoder_train.py -x -q -m gpu > gpurun_out/r05_z_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_z_tests.log
{
for rep in 1 2 3; do
  timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
} > gpurun_out/r05_z_ab.log 2>&1
( cd /tmp && rm -rf /tmp/prof_t5 && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
timeout 120 python tools/step_timeline.py $(find /tmp/prof_t5 -name "*results.db" | head -1) k_af_a 5 10 > gpurun_out/r05_config5_timeline_d.txt 2>&1
