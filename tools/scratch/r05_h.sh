#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "flash or hybrid or exophormer or bf16_mma" 2>&1 | tail -4
timeout 600 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision bf16 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('exo training bf16', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'])"
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_exo && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) 2>&1 | head -22 | cut -c1-150
for side in 6 8 12; do for pct in 10 30 60; do for m in auto force; do
  DA_HYBRID=$m timeout 300 python bench.py --config csr --side $side --pct $pct --puzzles 256 --steps 50 --warmup 5 --replays 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('side $side pct $pct G=256 DA_HYBRID=$m', round(d['value']), round(d['ms_per_step'],4), d['config']['attention_path'][:12])"
done; done; done
echo "== probe2 PIPELINED hidden G=32"; DA_OPT_HID=10 PROBE2=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 | tail -5
} > gpurun_out/r05_h.log 2>&1
