#!/bin/bash
# GPU box: the tail kernel's next-step embedding (DA_TAIL_NEXT=1) and the row-panel projections (DA_ENABLE_XPANEL=1) on the scripted small Batch: interleaved process pairs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
L=$O/r06_scripted_tail_next_ab.log; : > $L
run() { echo "$1 | $(env $1 timeout 300 python bench.py --config scripted --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), {k:round(v['ms_per_batch_step'],4) for k,v in t.items() if k in ('2','4')})")" >> $L; }
for i in 1 2 3 4; do run "DA_NOP=0"; run "DA_TAIL_NEXT=1"; run "DA_TAIL_NEXT=1 DA_ENABLE_XPANEL=1"; done
cat $L
