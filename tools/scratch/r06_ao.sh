#!/bin/bash
# GPU box: k_attn_tiny (K | V of a tiny complete graph in LDS, C = 104) -- parity, 3D suites, configuration 4 pairs (experiments build: DA_ATTN_TINY=0 = edge list)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiny or 3d or csr" > $O/r06_ao_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_ao_tests.log
L=$O/r06_config4_attn_tiny_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d['roofline'].get('classes',{}); print(round(d['ms_per_step'],4), {k:round(v['us_per_step'],1) for k,v in c.items()})")" >> $L; }
for i in 1 2 3 4; do run "DA_ATTN_TINY=0" "--config 4"; run "DA_ATTN_TINY=1" "--config 4"; done
cat $L
timeout 600 python bench.py --config 4 > $O/r06_bench_config_4.json 2>/dev/null; echo "bench rc=$?"
