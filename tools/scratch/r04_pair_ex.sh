#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
{
set -x
timeout 900 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_samplers.py -x -q -m gpu -k "two_branch or sampler or guided or cfg or ddpm or eta" 2>&1 | tail -5
timeout 600 python tools/pair_ex_probe.py 2>&1 | grep -E "two_branch|Error|error" 
} > gpurun_out/r04_pair_ex.log 2>&1
