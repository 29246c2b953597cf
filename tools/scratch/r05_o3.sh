#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
for rep in 1 2 3; do
for sm in 8 4; do
  DA_GEMM_SPLITK_MAX=$sm timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('config5 splitk_max=$sm bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
done
} > $O/r05_o3.log 2>&1
timeout -k 5 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bf16 or oracle or reference_fixture" > $O/r05_o3_tests.log 2>&1
echo "tests rc=$?" >> $O/r05_o3_tests.log
