#!/bin/bash
# GPU box: the virtual rows' kernel, four vs eight edges in flight (experiments build, interleaved process pairs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
L=$O/r06_virtual_rows_wide_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('two_batches_in_flight') or d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), 'two in flight per batch-step', round(t.get('ms_per_batch_step',0),4))")" >> $L; }
for i in 1 2 3 4 5; do for c in "--config 3" "--config 3 --degree 90"; do run "DA_CONT_WIDE=0" "$c"; run "DA_CONT_WIDE=1" "$c"; done; done
cat $L
