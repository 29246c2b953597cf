#!/bin/bash
# GPU box: virtual rows' conv-0 projections placed by the embedding's launch -- scripted / exophormer suites, then interleaved process pairs (experiments build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 2400 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_benched_mode.py tests/test_gpu_samplers.py tests/test_gpu_parity.py -m gpu -x -q > $O/r06_ag_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r06_ag_tests.log
L=$O/r06_virtual_scatter_in_embed_ab.log; : > $L
run() { echo "$1 | $2 | $(env DA_LIB_PATH=$EXP $1 timeout 300 python bench.py $2 --no-cpu-baseline --no-roofline --no-train-side --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); t=d.get('two_batches_in_flight') or d.get('batches_in_flight') or {}; print(round(d['ms_per_step'],4), {k:round(v['ms_per_batch_step'],4) for k,v in t.items() if k in ('2','4')} if 'batches_in_flight' in d else round(t.get('ms_per_batch_step',0),4))")" >> $L; }
for i in 1 2 3 4; do for c in "--config scripted" "--config 3"; do run "DA_VIRT_SCATTER_IN_EMBED=0" "$c"; run "DA_VIRT_SCATTER_IN_EMBED=1" "$c"; done; done
cat $L
