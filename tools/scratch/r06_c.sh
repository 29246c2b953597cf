#!/bin/bash
# GPU box: Batch-size sweep of the headline pair loop (two branches of G/2 puzzles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_batch_sweep.log; : > $L
run() { echo "puzzles $1 steps $2: $(timeout 200 python bench.py --puzzles $1 --steps $2 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)" >> $L; }
for rep in 1 2; do for G in 64 96 128 192 256; do run $G 20; done; done
for G in 64 128 256; do run $G 100; done
cat $L; tail -3 /tmp/err.log
