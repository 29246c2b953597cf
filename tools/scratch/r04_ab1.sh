# round 4, first A/B of the optimistic last-layer / masked kernels (run on the GPU box from the repo root)
set -x
python -m pytest tests/test_gpu_softmax_fallbacks.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_parity.py tests/test_gpu_benched_mode.py -x -q -k "forward_2d or rot900 or exo900 or conv_dense or config3 or two_branch" 2>&1 | tail -8
for rep in 1 2; do
  echo "== last layer, harness (G=64, n=900, C=144 folded)"
  DA_ATTN_OPT_LAST=1 tools/bin/attn_bench 64 900 144 1 50 1 0 1 2 | tail -2
  DA_ATTN_OPT_LAST=0 tools/bin/attn_bench 64 900 144 1 50 1 0 1 2 | tail -2
  DA_ATTN_OPT_LAST=0 DA_ATTN_LAST_FAST=0 tools/bin/attn_bench 64 900 144 1 50 1 0 1 2 | tail -2
  echo "== hidden layer, harness (G=64, n=900, C=32)"
  tools/bin/attn_bench 64 900 32 0 50 1 0 1 2 | tail -2
done
echo "== whole graph A/B (64 puzzles)"
for rep in 1 2 3; do
  for cfg in "DA_ATTN_OPT_LAST=1" "DA_ATTN_OPT_LAST=0" "DA_ATTN_OPT_LAST=0 DA_ATTN_LAST_FAST=0"; do
    env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('$cfg', round(d['value']), round(d['ms_per_step'],4))"
  done
done
echo "== config 3 A/B (32 puzzles, d=539)"
for rep in 1 2; do
  for cfg in "DA_ATTN_OPT_MASKED=1" "DA_ATTN_OPT_MASKED=0"; do
    env $cfg python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('$cfg', round(d['value']), round(d['ms_per_step'],4))"
  done
done
