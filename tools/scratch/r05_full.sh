#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r05_full_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_full_tests.log
{
for rep in 1 2 3; do
  timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
} > gpurun_out/r05_full_ab.log 2>&1
