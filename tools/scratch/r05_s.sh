#!/bin/bash
# two-slab optimistic kernels (da_attn_opt2.hip): correctness against the fp32 harness reference, then timing A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for sh in "4 900" "6 144" "7 150" "9 33" "5 161" "3 320" "2 1000"; do set -- $sh
  for v in 80; do
  echo "hid$v G=$1 n=$2"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench $1 $2 32 0 3 1 1 1 2 | grep -i "check\|rc\|fault"
  echo "last$v G=$1 n=$2"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench $1 $2 144 1 3 1 0 1 2 | grep -i "check\|rc\|fault"
  done
done
echo "hid80 sharp"; DA_OPT_HID=80 timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 40 2 | grep -i "check\|fault"
echo "last80 sharp"; DA_OPT_LAST=80 timeout 120 tools/bin/attn_bench 3 900 144 1 3 1 0 40 2 | grep -i "check\|fault"
echo "last81"; DA_OPT_LAST=81 timeout 120 tools/bin/attn_bench 3 900 144 1 3 1 0 1 2 | grep -i "check\|fault"
echo "hid83"; DA_OPT_HID=83 timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 1 2 | grep -i "check\|fault"
for rep in 1 2 3; do
  for v in 0 80 83; do
    echo "== hid v=$v G=32 n=900"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=64 n=900"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=512 n=144"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 512 144 32 0 50 0 0 1 2 | tail -1
  done
  for v in 0 80 81; do
    echo "== last v=$v G=32 n=900"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 32 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=64 n=900"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 64 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=512 n=144"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 512 144 144 1 50 0 0 1 2 | tail -1
  done
done
for rep in 1 2; do
  for cfg in "0 0" "80 0" "0 80" "80 80"; do
    set -- $cfg
    DA_OPT_HID=$1 DA_OPT_LAST=$2 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline hid=$1 last=$2', round(d['value']), round(d['ms_per_step'],4))"
  done
done
} > gpurun_out/r05_s.log 2>&1
