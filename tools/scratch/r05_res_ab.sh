#!/bin/bash
# whole-graph A/B of the K/V-resident hidden-layer attention (DA_ATTN_RES=1) against the ring kernel (0): headline (two-branch loop), config 3p at 64 puzzles one-branch if available
cd "$(dirname "$0")/.."
out=gpurun_out/r05_res_ab_graph.log
: > $out
for rep in 1 2 3; do
  for r in 0 1; do
    echo -n "RES=$r " >> $out
    DA_ATTN_RES=$r timeout 600 python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 > gpurun_out/r05_res_bench_${r}_${rep}.json
    python -c "import sys,json; d=json.load(open(sys.argv[1])); r=d.get('roofline',{}); print(d['value'], d['ms_per_step'], r.get('kernel'), r.get('frac'))" gpurun_out/r05_res_bench_${r}_${rep}.json >> $out 2>&1
  done
done
cat $out
