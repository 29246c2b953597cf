#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r05_o_tests.log
{
for rep in 1 2 3; do
  for v in 30 0; do
    DA_OPT_MASKED_VAR=$v timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('config3 d539 masked_var=$v', round(d['value']), round(d['ms_per_step'],4))"
    DA_OPT_MASKED_VAR=$v timeout 300 python bench.py --config 3 --degree 90 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('config3 d90 masked_var=$v', round(d['value']), round(d['ms_per_step'],4))"
  done
  for cfg in "30 30" "0 0"; do
    set -- $cfg
    DA_OPT_HID=$1 DA_OPT_LAST=$2 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline hid=$1 last=$2 (30 = old DMA form)', round(d['value']), round(d['ms_per_step'],4))"
  done
done
DA_OPT_HID=0 timeout 300 python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('config2 new', round(d['value']), round(d['ms_per_step'],4))"
DA_OPT_HID=30 DA_OPT_LAST=30 timeout 300 python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('config2 old DMA', round(d['value']), round(d['ms_per_step'],4))"
} > gpurun_out/r05_o.log 2>&1
