#!/bin/bash
# GPU box: timing-only probe -- what would Q / skip on the fly cost if x were stored fragment-major (1 KB contiguous per load instruction)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
L=$O/r06_qsf_fake_fm.log; : > $L
run() { echo "$1 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3; do run "DA_ATTN_RES_QSF=0"; run "DA_ATTN_RES_QSF=1"; run "DA_ATTN_RES_QSF=1 DA_QSF_FAKE_FM=1"; done
cat $L
