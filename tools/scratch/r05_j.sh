#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_samplers.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_benched_mode.py -x -q -k "training_curve or end_metric" -s 2>&1 | grep -i "training curve\|end metric\|passed\|failed\|error" | tail -8
} > gpurun_out/r05_j_tests.log 2>&1
ROUND=r05 bash tools/collect_round.sh > gpurun_out/r05_collect.log 2>&1
cp gpurun_out/r05_power_clock.log gpurun_out/r05_issue_probe.log profiles/r05/ 2>/dev/null
