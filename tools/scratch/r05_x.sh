#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/r05_x_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_x_tests.log
for p in bf16; do
timeout -k 5 240 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --precision $p 2>/dev/null | tail -1 > gpurun_out/r05_x_config5_$p.json
done
( cd /tmp && rm -rf /tmp/prof_t5 && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
timeout 120 python tools/step_timeline.py $(find /tmp/prof_t5 -name "*results.db" | head -1) k_af_a 5 10 > gpurun_out/r05_config5_timeline_c.txt 2>&1
