#!/bin/bash
# GPU box: the scripted Batch -- why do two Batches in flight not overlap?  (virtual rows on the caller's stream; step-rule switches forced)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_scripted_variants.log; : > $L
run() { echo "$1 | $(env $1 timeout 300 python bench.py --config scripted --no-cpu-baseline --no-roofline --no-train-side 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('two_batches_in_flight') or {}; print(round(d['ms_per_step'],4), 'two in flight per batch-step', round(t.get('ms_per_batch_step',0),4), 'x', round(t.get('vs_one_batch_in_flight',0),3))")" >> $L; }
for i in 1 2; do
run "DA_NONE=0"
run "DA_DISABLE_FOLDS=32"
run "DA_TAIL_NEXT=1"
run "DA_TAIL_NEXT=1 DA_DISABLE_FOLDS=32"
run "DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=1"
done
cat $L
