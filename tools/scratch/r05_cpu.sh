#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 300 python tools/train_cpu_probe.py > gpurun_out/r05_train_cpu_probe.log 2>&1
