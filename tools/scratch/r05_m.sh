#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for rep in 1 2; do
for dl in 0 80 160 240 320 480; do
  DA_PAIR_DELAY_US=$dl timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('headline delay $dl us', round(d['value']), round(d['ms_per_step'],4))"
done
done
DA_ENABLE_XPANEL=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('headline xpanel', round(d['value']), round(d['ms_per_step'],4))"
} > gpurun_out/r05_m.log 2>&1
