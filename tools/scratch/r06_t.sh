#!/bin/bash
# GPU box: Q and skip projected by the resident attention kernel (k_attn_res<.., 64>, DA_ATTN_RES_QSF, experiments build): bit-identity test, the
# 900-piece forward / trajectory parity with the switch on, then interleaved process pairs on the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 900 python -m pytest tests/test_gpu_attn_resident.py -m gpu -x -q -k "on_the_fly or projected or pyg_formula" > $O/r06_t_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_t_tests.log
DA_LIB_PATH=$EXP DA_ATTN_RES_QSF=1 timeout 1200 python -m pytest tests/test_gpu_benched_mode.py tests/test_gpu_attn_resident.py -m gpu -x -q -k "not subprocess" > $O/r06_t_tests_qsf_on.log 2>&1; echo "qsf-on suite rc=$?"; tail -5 $O/r06_t_tests_qsf_on.log
L=$O/r06_qsf_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3 4; do run "DA_ATTN_RES_QSF=0" "--steps 20 --warmup 5"; run "DA_ATTN_RES_QSF=1" "--steps 20 --warmup 5"; done
cat $L
