#!/bin/bash
# GPU box: the edge-list kernel with two destination rows per wave at C = 32 -- parity suites, then interleaved process pairs on --config csr / scripted
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 1500 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -x -q > $O/r06_r_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_r_tests.log
L=$O/r06_csr_two_rows_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), {k:round(v['ms_per_batch_step'],4) for k,v in t.items() if k in ('2','4')})")" >> $L; }
for i in 1 2 3 4 5; do for c in "--config csr" "--config scripted"; do run "DA_CSR_TWO_ROWS=0" "$c"; run "DA_CSR_TWO_ROWS=1" "$c"; done; done
cat $L
