#!/bin/bash
# wide workgroups (6 / 8 waves share one K / V stream) and the no-DMA timing ablations of the optimistic kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for v in 71 72 73; do
  echo "hid$v"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 1 2 | grep -i "check\|fault\|rc"
  echo "hid$v n=150"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 5 150 32 0 3 1 1 1 2 | grep -i "check\|fault\|rc"
done
for v in 71 72 73 74; do
  echo "last$v"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 3 900 144 1 3 1 0 1 2 | grep -i "check\|fault\|rc"
  echo "last$v n=150"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 5 150 144 1 3 1 1 1 2 | grep -i "check\|fault\|rc"
done
for rep in 1 2 3; do
  for v in 0 71 72 73; do
    echo "== hid v=$v G=32 n=900"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== hid v=$v G=64 n=900"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 0 0 1 2 | tail -1
  done
  for v in 0 71 72 73 74; do
    echo "== last v=$v G=32 n=900"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 32 900 144 1 50 0 0 1 2 | tail -1
    echo "== last v=$v G=64 n=900"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench 64 900 144 1 50 0 0 1 2 | tail -1
  done
  for v in 0 201 202; do
    echo "== ABL hid v=$v G=32"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench_abl 32 900 32 0 50 0 0 1 2 | tail -1
    echo "== ABL hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench_abl 64 900 32 0 50 0 0 1 2 | tail -1
    echo "== ABL last v=$v G=32"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench_abl 32 900 144 1 50 0 0 1 2 | tail -1
    echo "== ABL last v=$v G=64"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench_abl 64 900 144 1 50 0 0 1 2 | tail -1
  done
done
} > gpurun_out/r05_t.log 2>&1
