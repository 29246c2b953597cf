#!/bin/bash
# GPU box: where does round 2's one-kernel hidden conv (projection + attention, K / V projected straight into LDS) stand against today's two-kernel
# form in the two-graph step?  (experiments build, interleaved process pairs on the headline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
L=$O/r06_conv_fused_today.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3; do run "DA_CONV_FUSED=0" "--steps 20 --warmup 5"; run "DA_CONV_FUSED=1" "--steps 20 --warmup 5"; done
cat $L
