#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r05_i_tests.log
{
for m in auto off; do
  DA_HYBRID=$m timeout 600 python bench.py --config scripted --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('scripted DA_HYBRID=$m', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'][:12], d['training_step_same_batch'])"
done
timeout 600 python bench.py --config csr --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('csr default', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'], d['roofline']['kernel'], d['roofline']['frac'])"
DA_HYBRID=force timeout 600 python bench.py --config csr --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('csr forced hybrid', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'])"
} > gpurun_out/r05_i.log 2>&1
