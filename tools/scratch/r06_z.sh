#!/bin/bash
# GPU box: timing ablations of the pipelined key loop (generated asm variants; results wrong except PIPE = 0 / 1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
L=$O/r06_res_pipe_ablations.log; : > $L
run() { echo "$1 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items() if k.startswith('attn')})")" >> $L; }
for i in 1 2; do for v in 0 1 2 3 4 5 6; do run "DA_ATTN_RES_PIPE=$v"; done; done
cat $L
