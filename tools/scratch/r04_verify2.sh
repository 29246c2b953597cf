set -x
timeout 900 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_softmax_fallbacks.py tests/test_gpu_parity.py -x -q -k "exo or banded or expander or config3 or hybrid or forward_2d" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_encoder_train.py -x -q 2>&1 | tail -5
bash tools/r04_masked_probe.sh
for rep in 1 2; do
  for cfg in "DA_EXPANDER_LAYOUT=banded" "DA_EXPANDER_LAYOUT=natural"; do
    for deg in 539 90; do
    timeout 300 env $cfg python bench.py --config 3 --degree $deg --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('$cfg d=$deg', round(d['value']), round(d['ms_per_step'],4))"
    done
  done
done
for prec in fp32 bf16; do
  timeout 300 python bench.py --config 5 --precision $prec --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 $prec', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'])"
done
