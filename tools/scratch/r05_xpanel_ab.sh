#!/bin/bash
# GPU box: row-panel projections (DA_ENABLE_XPANEL=1) re-judged in the step under the two-graph pair loop; VERDICT r04 item 4c
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_xpanel_under_split_graphs.log; : > $L
run() { echo "$1 $2 $(env $1 $2 timeout 80 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode --replays 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3 4 5 6 7; do
  run DA_NOP=1 DA_NOP2=1; run DA_ENABLE_XPANEL=1 DA_NOP2=1; run DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=1
done
for i in 1 2 3; do run DA_NOP=1 "BENCH_CONFIG=2"; run DA_ENABLE_XPANEL=1 "BENCH_CONFIG=2"; done
cat $L
