#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
echo "== probe2 hidden G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 | tail -6
echo "== probe2 last G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 144 1 20 0 0 1 2 | tail -6
echo "== probe2 hidden G=64"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 64 900 32 0 20 0 0 1 2 | tail -6
} > gpurun_out/r05_k.log 2>&1
