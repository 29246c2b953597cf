#!/bin/bash
# GPU box: remainder-edge training kernels with U edges of a wave per trip -- training parity suites, exophormer training lines (bf16 operands / fp32), scripted training side
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_scripted.py -m gpu -x -q > $O/r06_al_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_al_tests.log
for i in 1 2; do
timeout 900 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('exo bf16', round(d['ms_per_step'],4))"
timeout 900 python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision fp32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('exo fp32', round(d['ms_per_step'],4))"
done
( cd /tmp && rm -rf /tmp/prof_exo && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) > $O/r06_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt 2>&1
grep "k_attn_irr" $O/r06_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt | cut -c1-140
