#!/bin/bash
# GPU box: DA_STEP_AUTO on / off at the driver's own bench arguments (--steps 20 --warmup 5)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_step_auto_ab_steps20.log; : > $L
run() { echo "$1 $(env $1 timeout 80 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'], d.get('replay_median_ms'))")" >> $L; }
for i in 1 2 3 4 5 6; do run DA_STEP_AUTO=1; run DA_STEP_AUTO=0; done
cat $L
