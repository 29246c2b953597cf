#!/bin/bash
# GPU box: which regime decides -- 31 back-to-back timed passes (bench default, the driver's line) against one pass after idle (--replays 0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_step_auto_ab_regimes.log; : > $L
run() { echo "$1 $2 | $3 | $(env $1 $2 timeout 80 python bench.py $3 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3; do
  run DA_STEP_AUTO=0 DA_TAIL_NEXT=1 "--steps 20 --warmup 5"; run DA_STEP_AUTO=0 X=1 "--steps 20 --warmup 5"
  run DA_STEP_AUTO=1 X=1 "--steps 100 --warmup 10"; run DA_STEP_AUTO=0 X=1 "--steps 100 --warmup 10"
done
cat $L
