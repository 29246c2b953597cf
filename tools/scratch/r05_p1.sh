#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests/test_gpu_pcd_encoder.py tests/test_gpu_train.py -x -q -m gpu -k "pcd or side_stream or encoder or knn or nearest" > gpurun_out/r05_p1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_p1_tests.log
{
for rep in 1 2 3; do
for two in 1 0; do
  DA_PCD_TWO_STREAMS=$two timeout -k 5 300 python bench.py --mode encode --config 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('pcd encode two_streams=$two', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms')"
done
done
} > gpurun_out/r05_p1_ab.log 2>&1
