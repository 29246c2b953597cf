#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
echo "== probe2 hidden G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 | tail -3
echo "== probe2 last G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 144 1 20 0 0 1 2 | tail -3
echo "== probe2 hidden G=64"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 64 900 32 0 20 0 0 1 2 | tail -3
echo "== probe2 hidden G=4 (one round, quarter-full chip)"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 4 900 32 0 20 0 0 1 2 | tail -3
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -5
for m in auto force; do
  DA_HYBRID=$m timeout 600 python bench.py --config csr --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('csr G=512 DA_HYBRID=$m', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'])"
  DA_HYBRID=$m timeout 600 python bench.py --config csr --puzzles 64 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('csr G=64 DA_HYBRID=$m', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'])"
  DA_HYBRID=$m timeout 600 python bench.py --config scripted --steps 60 --warmup 5 --no-cpu-baseline --no-train-side 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('scripted DA_HYBRID=$m', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'])"
done
} > gpurun_out/r05_f.log 2>&1
