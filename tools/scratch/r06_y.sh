#!/bin/bash
# GPU box: software-pipelined key loop of the resident kernel (generated asm, DA_ATTN_RES_PIPE=1, experiments build): bit-identity test, 900-piece suites with it on, process pairs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 900 python -m pytest tests/test_gpu_attn_resident.py -m gpu -x -q -k "pipelined" > $O/r06_y_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_y_tests.log
DA_LIB_PATH=$EXP DA_ATTN_RES_PIPE=1 timeout 1200 python -m pytest tests/test_gpu_attn_resident.py tests/test_gpu_benched_mode.py -m gpu -x -q -k "not subprocess" > $O/r06_y_tests_on.log 2>&1; echo "pipe-on suite rc=$?"; tail -3 $O/r06_y_tests_on.log
L=$O/${TAG:-r06_res_pipe_ab1}.log; : > $L
run() { echo "$1 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3 4; do run "DA_ATTN_RES_PIPE=0"; run "DA_ATTN_RES_PIPE=1"; done
cat $L
