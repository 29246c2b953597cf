#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15
for p in bf16 fp32; do
  timeout 600 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision $p --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('exo training $p', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'])"
done
DA_HYB_FLASH=0 timeout 600 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision bf16 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('exo training bf16 pairs route', round(d['value']), round(d['ms_per_step'],3))"
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_exo && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) 2>&1 | head -30 | cut -c1-160
echo "== probe2 hidden G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 | tail -5
echo "== probe2 hidden G=4"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 4 900 32 0 20 0 0 1 2 | tail -5
echo "== probe2 last G=32"; PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 144 1 20 0 0 1 2 | tail -5
} > gpurun_out/r05_g.log 2>&1
