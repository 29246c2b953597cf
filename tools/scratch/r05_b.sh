#!/bin/bash
# round 5, batch B: timing ablations of the two optimistic kernels (what is the time sensitive to?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in 0 4 104 108 116 132 124 128 160; do
  echo "== last v=$v G=64"; DA_OPT_LAST=$v timeout 120 tools/bin/attn_bench_ablate 64 900 144 1 30 0 0 1 2 | tail -1
done
for v in 0 1 104 108 116 132 160; do
  echo "== hid v=$v G=64"; DA_OPT_HID=$v timeout 120 tools/bin/attn_bench_ablate 64 900 32 0 30 0 0 1 2 | tail -1
done
echo "== hid v=0 no GELU G=64"; ACT_NONE=1 timeout 120 tools/bin/attn_bench_ablate 64 900 32 0 30 0 0 1 2 | tail -1
echo "== hid v=160 no GELU G=64"; ACT_NONE=1 DA_OPT_HID=160 timeout 120 tools/bin/attn_bench_ablate 64 900 32 0 30 0 0 1 2 | tail -1
done
echo "== check adds variants"
DA_OPT_LAST=4 timeout 120 tools/bin/attn_bench_ablate 4 900 144 1 3 1 0 1 2 | grep check
DA_OPT_HID=1 timeout 120 tools/bin/attn_bench_ablate 4 900 32 0 3 1 0 1 2 | grep check
} > gpurun_out/r05_b.log 2>&1
