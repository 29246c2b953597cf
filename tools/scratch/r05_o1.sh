#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_encoder_train.py -x -q -m gpu -k "adafactor or optimizer or training_step or data_parallel or ddp or curve" > $O/r05_o1_tests.log 2>&1
echo "tests rc=$?" >> $O/r05_o1_tests.log
{
for rep in 1 2 3; do
  timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4), d['roofline']['phase_share'])"
done
timeout -k 5 300 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('exo 16x900 bf16', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
} > $O/r05_o1_ab.log 2>&1
