#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
DA_OPT_HID=40 timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep check
DA_OPT_LAST=3 timeout 120 tools/bin/attn_bench 4 900 144 1 3 1 0 1 2 | grep check
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    echo "== last v=$v G=32"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 32 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=64"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 64 900 144 1 50 0 0 1 2 | tail -1
  done
  for v in 0 40; do
    echo "== hid v=$v G=32"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
  done
done
for rep in 1 2 3; do
  for cfg in "0 0" "0 3" "40 0" "0 2"; do
    set -- $cfg
    DA_OPT_HID=$1 DA_OPT_LAST=$2 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline hid=$1 last=$2', round(d['value']), round(d['ms_per_step'],4))"
  done
done
} > gpurun_out/r05_p.log 2>&1
