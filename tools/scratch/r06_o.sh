#!/bin/bash
# GPU box: virtual-rows kernel with packed rows / eight edges in flight: hybrid parity + configuration 3 (both degrees) + scripted
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_scripted.py tests/test_gpu_softmax_fallbacks.py -m gpu -x -q 2>&1 | tail -2
L=$O/r06_virtual_rows_ab.log; : > $L
run() { echo "$1 | $(timeout 300 python bench.py $1 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('two_batches_in_flight') or d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), 'two in flight per batch-step', round(t.get('ms_per_batch_step',0),4))")" >> $L; }
for i in 1 2 3; do run "--config 3"; run "--config 3 --degree 90"; run "--config scripted"; done
cat $L
