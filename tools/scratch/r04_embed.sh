#!/bin/bash
# k_embed_mlp0: parity tests of the bf16 inference path + whole-graph A/B (headline, config 2, config 3)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
set -x
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benched_mode.py tests/test_gpu_samplers.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
for cfg in "DA_EMBED_FUSED=1" "DA_EMBED_FUSED=0"; do
  env $cfg timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('headline $cfg', round(d['value']), round(d['ms_per_step'],4))"
done
done
for cfg in "DA_EMBED_FUSED=1" "DA_EMBED_FUSED=0"; do
  env $cfg timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config3 $cfg', round(d['value']), round(d['ms_per_step'],4))"
  env $cfg timeout 300 python bench.py --config 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config2 $cfg', round(d['value']), round(d['ms_per_step'],4))"
  env $cfg timeout 300 python bench.py --config 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config4 $cfg', round(d['value']), round(d['ms_per_step'],4))"
done
} > gpurun_out/r04_embed.log 2>&1
