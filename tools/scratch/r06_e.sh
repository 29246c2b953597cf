#!/bin/bash
# GPU box, round 6 call 5: edge-list kernel with several edges in flight, two exophormer Batches in flight, progressive landing of k_attn_res
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 1200 python -m pytest tests/test_gpu_scripted.py tests/test_gpu_parity.py -m gpu -x -q > $O/r06_e_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/r06_e_tests.log
# progressive landing: parity (the resident kernel's suite on the experiments build with the variant selected) and timing
DA_LIB_PATH=$EXP DA_ATTN_RES_PH=8 timeout 900 python -m pytest tests/test_gpu_attn_resident.py -m gpu -x -q > $O/r06_e_prog_parity.log 2>&1; echo "prog parity rc=$?"; tail -3 $O/r06_e_prog_parity.log
L=$O/r06_progressive_landing.log; : > $L
for G in 32 64; do for PH in 1 8; do echo "G=$G DA_ATTN_RES_PH=$PH: $(DA_ATTN_RES_PH=$PH timeout 120 tools/bin/attn_bench $G 900 32 0 400 1 0 1 2 2>&1 | tail -2 | tr '\n' ' ')" >> $L; done; done
run() { echo "$1 $(env DA_LIB_PATH=$EXP $1 timeout 120 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3 4 5 6; do run DA_ATTN_RES_PH=1 20; run DA_ATTN_RES_PH=8 20; done
echo "--- steps 100" >> $L
for i in 1 2 3; do run DA_ATTN_RES_PH=1 100; run DA_ATTN_RES_PH=8 100; done
cat $L
timeout 900 python bench.py --config csr --no-cpu-baseline > $O/r06_bench_csr_quick.json 2>$O/r06_bench_csr_quick.err; python -c "
import json; d=json.loads([l for l in open('$O/r06_bench_csr_quick.json') if l.startswith('{')][-1]); print('csr', d['ms_per_step'], d['value'], json.dumps(d.get('roofline',{}).get('sparse_path', d.get('roofline',{})))[:600])"
timeout 900 python bench.py --config 3 --no-cpu-baseline --no-roofline --no-parity-mode > $O/r06_bench_config3_quick.json 2>$O/r06_bench_config3_quick.err; python -c "
import json; d=json.loads([l for l in open('$O/r06_bench_config3_quick.json') if l.startswith('{')][-1]); print('config3', d['ms_per_step'], d['value'], d.get('two_batches_in_flight'))"; tail -3 $O/r06_bench_config3_quick.err
timeout 900 python bench.py --config 3 --degree 90 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('config3 d90', d['ms_per_step'], d['value'], d.get('two_batches_in_flight'))"
