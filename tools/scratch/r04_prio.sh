#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  echo "== base hidden";  timeout 120 tools/bin/attn_bench 64 900 32 0 50 1 0 1 2 | tail -1
  echo "== prio hidden";  LD_LIBRARY_PATH=$PWD/tools/bin/prio:$LD_LIBRARY_PATH timeout 120 tools/bin/attn_bench 64 900 32 0 50 1 0 1 2 | tail -1
  echo "== base last";  timeout 120 tools/bin/attn_bench 64 900 144 1 50 1 0 1 2 | tail -1
  echo "== prio last";  LD_LIBRARY_PATH=$PWD/tools/bin/prio:$LD_LIBRARY_PATH timeout 120 tools/bin/attn_bench 64 900 144 1 50 1 0 1 2 | tail -1
  echo "== base hidden half";  timeout 120 tools/bin/attn_bench 32 900 32 0 50 1 0 1 2 | tail -1
  echo "== prio hidden half";  LD_LIBRARY_PATH=$PWD/tools/bin/prio:$LD_LIBRARY_PATH timeout 120 tools/bin/attn_bench 32 900 32 0 50 1 0 1 2 | tail -1
done
} > gpurun_out/r04_prio.log 2>&1
