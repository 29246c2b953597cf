#!/bin/bash
# round 5, batch A: last-layer / hidden-layer kernel variants in the harness, -fno-slp-vectorize on da_attn_opt.hip, headline baseline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
echo "== correctness of every variant (G=4)"
for v in 0 1 2 3 4 5 6 7; do echo "last v=$v"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 4 900 144 1 3 1 0 1 2 | grep check; done
for v in 0 1 2 3; do echo "hid v=$v"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 4 900 32 0 3 1 0 1 2 | grep check; done
for rep in 1 2; do
  for v in 0 1 2 3 4 5 6 7; do
    echo "== last v=$v G=64"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 64 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=32"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 32 900 144 1 50 0 0 1 2 | tail -1
  done
  for v in 0 1 2 3; do
    echo "== hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=32"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
  done
  echo "== noslp last G=64"; timeout 120 tools/bin/attn_bench_noslp 64 900 144 1 50 0 0 1 2 | tail -1
  echo "== noslp last G=32"; timeout 120 tools/bin/attn_bench_noslp 32 900 144 1 50 0 0 1 2 | tail -1
  echo "== noslp hid G=64"; timeout 120 tools/bin/attn_bench_noslp 64 900 32 0 50 0 0 1 2 | tail -1
  echo "== noslp hid G=32"; timeout 120 tools/bin/attn_bench_noslp 32 900 32 0 50 0 0 1 2 | tail -1
done
for rep in 1 2 3; do
  for lib in lib lib_noslp; do
    DA_LIB_PATH=$PWD/diffassemble_amd/$lib/libdiffassemble_hip.so timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline $lib', round(d['value']), round(d['ms_per_step'],4))"
  done
done
for lib in lib lib_noslp; do
  DA_LIB_PATH=$PWD/diffassemble_amd/$lib/libdiffassemble_hip.so timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config3 $lib', round(d['value']), round(d['ms_per_step'],4))"
done
} > gpurun_out/r05_a.log 2>&1
