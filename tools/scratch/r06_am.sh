#!/bin/bash
# GPU box: 20-piece graphs (BASELINE configuration 4, 3D) -- block-diagonal MFMA attention (default) against the edge-list kernels (DA_DISABLE_DENSE=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_config4_dense_vs_csr.log; : > $L
run() { echo "$1 | $2 | $(env $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3; do run "DA_NOP=1" "--config 4"; run "DA_DISABLE_DENSE=1" "--config 4"; done
run "DA_NOP=1" "--config 1"; run "DA_DISABLE_DENSE=1" "--config 1"
cat $L
