#!/bin/bash
# GPU box: the round's "faster alone, not in the step" variants re-judged under the two-graph pair loop (DA_PAIR_SPLIT=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_rejudge_under_split_graphs.log; : > $L
run() { echo "$1 $2 $(env $1 timeout 80 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode --replays 0 $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3; do
  for v in DA_NOP=1 DA_TAIL_NEXT=1 DA_ENABLE_XPANEL=1 DA_ATTN_RES=0 DA_CONV_FUSED=1 DA_ATTN2=1; do run $v ""; done
  run DA_NOP=1 "--puzzles 128"; run DA_NOP=1 "--puzzles 96"
done
cat $L
