#!/bin/bash
# round 5, batch C: software-pipelined optimistic pass (VAR bit 64) in the harness
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
echo "== correctness"
for v in 10 11 12; do echo "last v=$v"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 4 900 144 1 3 1 0 1 2 | grep -i "check\|bad\|rc"; done
for v in 10 11 12; do echo "hid v=$v"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep -i "check\|bad\|rc"; done
for v in 10; do echo "hid v=$v nodiag n=150"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 7 150 32 0 3 1 1 1 2 | grep -i "check\|bad\|rc"; done
for v in 10; do echo "last v=$v nodiag n=33"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 7 33 144 1 3 1 1 1 2 | grep -i "check\|bad\|rc"; done
for v in 10; do echo "last v=$v sharp (fallback) n=900"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 2 900 144 1 3 1 0 40 2 | grep -i "check\|bad\|rc"; done
for rep in 1 2; do
  for v in 0 10 11 12; do
    echo "== last v=$v G=64"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 64 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=32"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 32 900 144 1 50 0 0 1 2 | tail -1
  done
  for v in 0 10 11 12; do
    echo "== hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=32"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
  done
done
for rep in 1 2; do
  for cfg in "0 0" "10 10" "10 0" "0 10"; do
    set -- $cfg
    DA_OPT_HID=$1 DA_OPT_LAST=$2 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline hid=$1 last=$2', round(d['value']), round(d['ms_per_step'],4))"
  done
done
} > gpurun_out/r05_c.log 2>&1
