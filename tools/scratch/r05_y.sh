#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r05_y_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_y_tests.log
{
for rep in 1 2; do
for sdw in 1 0; do
  DA_TRAIN_SIDE_DW=$sdw timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16 side_dw=$sdw', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
done
for sdw in 1 0; do
  DA_TRAIN_SIDE_DW=$sdw timeout -k 5 300 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('exo bf16 side_dw=$sdw', round(d['ms_per_step'],4), 'fp32', round(d['fp32_reference_arithmetic']['ms_per_step'],4))"
done
} > gpurun_out/r05_y_ab.log 2>&1
