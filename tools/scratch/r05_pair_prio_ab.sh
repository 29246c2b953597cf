#!/bin/bash
# GPU box: pair-loop experiments of the round's last session -- pair-stream priority (DA_PAIR_PRIO) and unequal branches (DA_PAIR_SPLIT_AT)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_pair_prio_split_at_ab.log; : > $L
run() { echo "$1 $(env $1 timeout 60 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode --replays 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3; do for v in DA_PAIR_PRIO=0 DA_PAIR_PRIO=1 DA_PAIR_PRIO=-1 DA_PAIR_SPLIT_AT=36 DA_PAIR_SPLIT_AT=40 DA_PAIR_SPLIT_AT=24; do run $v; done; done
cat $L
