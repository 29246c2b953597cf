#!/bin/bash
# GPU box: hybrid graphs -- internal side-stream forks vs two Batches in flight (configuration 3, scripted)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_side_stream_vs_in_flight.log; : > $L
run() { echo "$1 | $2 | $(env $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('two_batches_in_flight') or {}; print(round(d['ms_per_step'],4), 'two in flight per batch-step', round(t.get('ms_per_batch_step',0),4), 'x', round(t.get('vs_one_batch_in_flight',0),3))")" >> $L; }
for i in 1 2; do
run "DA_NONE=0" "--config scripted"
run "DA_NONE=0" "--config 3"
run "DA_DISABLE_FOLDS=32" "--config 3"
run "DA_NONE=0" "--config 3 --degree 90"
run "DA_DISABLE_FOLDS=32" "--config 3 --degree 90"
done
cat $L
timeout 600 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_samplers.py -m gpu -x -q 2>&1 | tail -2
