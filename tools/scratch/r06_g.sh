#!/bin/bash
# GPU box, round 6 call 7: the sweep's four candidates against the baseline, 8 interleaved process pairs each; new tests (capture, loss glue)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
timeout 900 python -m pytest tests/test_gpu_samplers.py tests/test_gpu_train.py -m gpu -x -q -k "capture or glue or p_losses or loss" > $O/r06_g_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r06_g_tests.log
L=$O/r06_experiment_candidates.log; : > $L
run() { echo "$1 | $(env DA_LIB_PATH=$EXP $1 timeout 120 python bench.py --steps 20 --warmup 5 --replays 40 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'])")" >> $L; }
for V in "DA_OPT_LAST=6" "DA_OPT_LAST=4" "DA_PAIR_PRIO=-1" "DA_EMBED_NPW=4"; do
  for i in 1 2 3 4 5 6 7 8; do run "DA_NONE=0"; run "$V"; done
done
python - <<'PY'
import math, statistics
rows=[l.strip().split(' | ') for l in open('gpurun_out/r06_experiment_candidates.log') if ' | ' in l]
i=0
while i < len(rows):
    tag=rows[i+1][0]; a=[]; b=[]
    while i+1 < len(rows) and rows[i+1][0]==tag:
        a.append(float(rows[i][1])); b.append(float(rows[i+1][1])); i+=2
    d=[y-x for x,y in zip(a,b)]
    pos,neg=sum(x>0 for x in d),sum(x<0 for x in d); n=pos+neg; k=min(pos,neg)
    p=min(1.0, 2*sum(math.comb(n,j) for j in range(k+1))/2**n) if n else 1.0
    print(f"{tag:20s} baseline median {statistics.median(a):.4f}  variant {statistics.median(b):.4f}  median diff {1e3*statistics.median(d):+.1f} us ({100*statistics.median(d)/statistics.median(a):+.2f} %)  variant faster in {neg} of {n}, sign test p = {p:.3f}")
PY
