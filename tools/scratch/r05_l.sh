#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
echo "== correctness"; timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep check; DA_OPT_HID=20 timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep check; DA_OPT_HID=21 timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep check
timeout 120 tools/bin/attn_bench 7 150 32 0 3 1 1 1 2 | grep check
timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 40 2 | grep check
for rep in 1 2 3; do
  for b in attn_bench_prev attn_bench; do
    echo "== $b hid G=32"; timeout 120 tools/bin/$b 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== $b hid G=64"; timeout 120 tools/bin/$b 64 900 32 0 50 0 0 1 2 | tail -1
  done
  for v in 20 21; do
    echo "== hid v=$v G=32"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
  done
done
for rep in 1 2 3; do
  for lib in lib_prev lib; do
    DA_LIB_PATH=$PWD/diffassemble_amd/$lib/libdiffassemble_hip.so timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline $lib', round(d['value']), round(d['ms_per_step'],4))"
  done
  DA_OPT_HID=20 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('headline hid=20', round(d['value']), round(d['ms_per_step'],4))"
done
} > gpurun_out/r05_l.log 2>&1
