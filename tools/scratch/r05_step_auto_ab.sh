#!/bin/bash
# GPU box: the four combinations of row-panel projections x next-step embedding in the tail kernel under the two-graph pair loop, the unset
# default (DA_STEP_AUTO: both for >= 512-piece graphs), configuration 2 with and without the rule; then the parity tests that run such loops
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; L=gpurun_out/r05_step_auto_ab.log; : > $L
run() { echo "$1 $2 $3 $(env $1 $2 $3 timeout 80 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode --replays 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3 4 5; do
  run DA_ENABLE_XPANEL=0 DA_TAIL_NEXT=0 X=1; run DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=0 X=1; run DA_ENABLE_XPANEL=0 DA_TAIL_NEXT=1 X=1; run DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=1 X=1; run DEFAULT=1 X=1 Y=1
done
for i in 1 2; do run DEFAULT=1 BENCH_CONFIG=2 X=1; run DA_STEP_AUTO=0 BENCH_CONFIG=2 X=1; done
cat $L
(time timeout 225 python -m pytest tests/test_gpu_tail_next.py tests/test_gpu_benched_mode.py tests/test_gpu_parity.py -x -q -k "tail or rot900 or drift_vs_fp32 or two_branch or row_panel or deterministic_under_load") > gpurun_out/r05_step_auto_tests.log 2>&1
tail -6 gpurun_out/r05_step_auto_tests.log
