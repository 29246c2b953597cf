#!/bin/bash
# hidden-layer optimistic kernel: five workgroups per CU (<= 96 VGPRs, three ring stages) against four (harness + whole graph)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
for rep in 1 2; do
  for v in 0 1; do
    echo "== harness W5=$v"; DA_ATTN_OPT_HID_W5=$v timeout 120 tools/bin/attn_bench 64 900 32 0 50 1 0 1 2 | tail -2
    echo "== harness half batch W5=$v"; DA_ATTN_OPT_HID_W5=$v timeout 120 tools/bin/attn_bench 32 900 32 0 50 1 0 1 2 | tail -1
  done
done
for rep in 1 2 3; do
  for v in 0 1; do
    DA_ATTN_OPT_HID_W5=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('headline W5=$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done
} > gpurun_out/r04_w5.log 2>&1
