#!/bin/bash
# GPU box, round 6 call 6: sweep of the experiment switches on the headline step (experiments build, one process per run, baseline interleaved)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; EXP=$GRAFT_REPO_ROOT/diffassemble_amd/lib_exp/libdiffassemble_hip.so
L=$O/r06_experiment_sweep.log; : > $L
run() { echo "$1 | $(env DA_LIB_PATH=$EXP $1 timeout 120 python bench.py --steps 20 --warmup 5 --replays 40 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'])")" >> $L; }
for V in "DA_ATTN_DUAL=1" "DA_OPT_LAST=1" "DA_OPT_LAST=2" "DA_OPT_LAST=3" "DA_OPT_LAST=4" "DA_OPT_LAST=6" "DA_OPT_LAST=10" "DA_OPT_LAST=11" "DA_OPT_LAST=60" "DA_OPT_LAST=70" \
         "DA_ATTN_RES_PH=3" "DA_ATTN_RES_PH=4" "DA_ATTN_RES_PH=5" "DA_ATTN_RES_PH=6" "DA_ATTN_RES_PH=7" "DA_PAIR_PRIO=1" "DA_PAIR_PRIO=-1" "DA_EMBED_NPW=4" "DA_EMBED_NPW=16" \
         "DA_ENABLE_XPANEL=0 DA_WREG_DIRECT=1" "DA_ENABLE_XPANEL=0 DA_WREG2=1" "DA_ENABLE_XPANEL=0 DA_WREG2=0"; do
  run "DA_NONE=0"; run "$V"; run "DA_NONE=0"; run "$V"
done
python - <<'PY'
import collections, statistics
rows=[l.strip().split(' | ') for l in open('gpurun_out/r06_experiment_sweep.log') if ' | ' in l]
base=[float(v) for k,v in rows if k=='DA_NONE=0' and v]
print(f"baseline: n={len(base)} median {statistics.median(base):.4f} min {min(base):.4f} max {max(base):.4f}")
d=collections.defaultdict(list)
for k,v in rows:
    if k!='DA_NONE=0' and v: d[k].append(float(v))
for k,v in d.items(): print(f"{k:42s} {statistics.mean(v):.4f}  ({100*(statistics.mean(v)/statistics.median(base)-1):+.2f} %)  {v}")
PY
