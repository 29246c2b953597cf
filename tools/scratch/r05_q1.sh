#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 1500 python -m pytest tests/test_gpu_pcd_encoder.py tests/test_gpu_train.py -x -q -m gpu -k "pcd or side_stream or encoder or knn or nearest" > gpurun_out/r05_q1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_q1_tests.log
timeout -k 5 300 python bench.py --mode encode --config 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_q1_pcd.json
