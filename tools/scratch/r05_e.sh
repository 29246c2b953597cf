#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 tools/bin/issue_probe > gpurun_out/r05_issue_probe.log 2>&1
timeout 300 python tools/scratch/r05_dbg_stage.py > gpurun_out/r05_dbg_stage.log 2>&1
{
timeout 900 python -m pytest tests/test_gpu_scripted.py -x -q 2>&1 | tail -15
timeout 600 python bench.py --config scripted --steps 60 --warmup 5 > gpurun_out/r05_bench_scripted.json 2> gpurun_out/r05_bench_scripted.err; tail -3 gpurun_out/r05_bench_scripted.err
timeout 600 python bench.py --config csr --steps 50 --warmup 5 > gpurun_out/r05_bench_csr.json 2> gpurun_out/r05_bench_csr.err; tail -3 gpurun_out/r05_bench_csr.err
} > gpurun_out/r05_e.log 2>&1
