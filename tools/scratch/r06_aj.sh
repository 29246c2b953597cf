#!/bin/bash
# GPU box: the tail kernel's next-step embedding with the pose MLP on the matrix cores (exact fp32): tail-next tests, then interleaved process pairs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 1500 python -m pytest tests/test_gpu_tail_next.py tests/test_gpu_benched_mode.py tests/test_gpu_samplers.py -m gpu -x -q > $O/r06_aj_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_aj_tests.log
L=$O/r06_tail_next_pos_mfma_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), {k:round(v['ms_per_batch_step'],4) for k,v in t.items() if k in ('2','4')})")" >> $L; }
for i in 1 2 3 4 5; do run "DA_TAIL_NEXT_POS_MFMA=0" "--steps 20 --warmup 5"; run "DA_TAIL_NEXT_POS_MFMA=1" "--steps 20 --warmup 5"; done
for i in 1 2 3; do run "DA_NOP=1" "--config scripted"; run "DA_TAIL_NEXT=1 DA_TAIL_NEXT_POS_MFMA=0" "--config scripted"; run "DA_TAIL_NEXT=1 DA_TAIL_NEXT_POS_MFMA=1" "--config scripted"; done
for i in 1 2 3; do run "DA_TAIL_NEXT_POS_MFMA=0" "--config 2"; run "DA_TAIL_NEXT_POS_MFMA=1" "--config 2"; done
cat $L
