#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
for rep in 1 2 3 4 5 6; do
  timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline > /tmp/o.json 2> /tmp/o.err
  echo "rc=$? $(tail -c 300 /tmp/o.err | tr '\n' ' ')"
  python -c "import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('config5 bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
} > $O/r05_o2.log 2>&1
