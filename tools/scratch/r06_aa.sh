#!/bin/bash
# GPU box: k_loss_grad with eight loads in flight -- glue test, config 5 line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "p_losses or glue or loss" > $O/r06_aa_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_aa_tests.log
for i in 1 2; do timeout 600 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['ms_per_step'],4), d['phases_ms'])"; done
