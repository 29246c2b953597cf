#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "exophormer or hybrid or flash or side_stream" > $O/r05_h1_tests.log 2>&1
echo "tests rc=$?" >> $O/r05_h1_tests.log
{
for rep in 1 2; do
for hs in 1 0; do
  DA_HYB_SIDE=$hs timeout -k 5 300 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('exo 16x900 hyb_side=$hs bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
  DA_HYB_SIDE=$hs timeout -k 5 300 python bench.py --config scripted --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('scripted hyb_side=$hs sampling ms', round(d['ms_per_step'],4), 'train', d['training_step_same_batch'])"
done
done
} > $O/r05_h1_ab.log 2>&1
