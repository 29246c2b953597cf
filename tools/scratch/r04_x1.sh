#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "bf16" 2>&1 | tail -3
for rep in 1 2 3; do
for cfg in "DA_ATTN_SMALL_ALL4=1" "DA_ATTN_SMALL_ALL4=0"; do
  env $cfg timeout 300 python bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 $cfg', round(d['value']), round(d['ms_per_step'],4))"
done
done
bash tools/r04_nst3.sh; cat gpurun_out/r04_nst3.log
} > gpurun_out/r04_x1.log 2>&1
