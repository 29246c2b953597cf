#!/bin/bash
# GPU box, round 6 call 2: where the last-layer projection of the thin kernel differs; concurrent timelines of the headline pair loop
# (default, thin everywhere, thin on the 1024-column projections only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 tools/bin/corun_probe 32 900 20 2>&1 | head -30 > $O/r06_corun_probe_b.log; cat $O/r06_corun_probe_b.log
for V in 0 1 2; do
  ( cd /tmp && rm -rf /tmp/prof_tl$V && DA_GEMM_THIN=$V timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_tl$V -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode --replays 3 > /tmp/prof_tl$V.log 2>&1 )
  DB=$(find /tmp/prof_tl$V -name "*results.db" | head -1)
  echo "== DA_GEMM_THIN=$V" > $O/r06_pair_timeline_thin$V.txt
  python tools/pair_timeline.py $DB --steps 20 --dump 60 >> $O/r06_pair_timeline_thin$V.txt 2>&1
  head -14 $O/r06_pair_timeline_thin$V.txt
done
L=$O/r06_thin2_ab_steps20.log; : > $L
run() { echo "$1 $(env $1 timeout 120 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3; do run DA_GEMM_THIN=0 20; run DA_GEMM_THIN=2 20; done
cat $L
