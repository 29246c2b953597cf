#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
} > gpurun_out/r05_d_tests.log 2>&1
bash tools/power_clock_probe.sh > gpurun_out/r05_power_clock.log 2>&1
