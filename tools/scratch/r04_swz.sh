set -x
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benched_mode.py tests/test_gpu_softmax_fallbacks.py -x -q -k "forward_2d or rot900 or exo or banded or conv_dense or hybrid or logit_offsets or two_branch or config3" 2>&1 | tail -5
for rep in 1 2; do
  tools/bin/attn_bench 64 900 32 0 50 1 0 1 2 | tail -2
  tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
done
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('headline', round(d['value']), round(d['ms_per_step'],4))"
  for deg in 539 90; do
    timeout 300 python bench.py --config 3 --degree $deg --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('config3 d=$deg', round(d['value']), round(d['ms_per_step'],4))"
  done
done
