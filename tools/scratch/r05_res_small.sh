#!/bin/bash
# k_attn_res<5>: 12 x 12 puzzles (144 pieces) -- parity, harness timing against the ring kernel, configuration 2 A/B
cd "$(dirname "$0")/.."
out=gpurun_out/r05_res_small.log
: > $out
for sm in 1 2; do
  echo "== DA_ATTN_RES_SMALL=$sm parity (n = 144, 129, 160, 150; with / without diagonal; sharp)" >> $out
  for n in 144 129 160 150; do for nd in 0 1; do DA_ATTN_RES_SMALL=$sm timeout 120 tools/bin/attn_bench 7 $n 32 0 3 1 $nd 1 2 2>&1 | grep -E "check" >> $out; done; done
  DA_ATTN_RES_SMALL=$sm timeout 120 tools/bin/attn_bench 7 144 32 0 3 1 0 40 2 2>&1 | grep -E "check" >> $out
  DA_ATTN_RES_SMALL=$sm DA_ATTN_FORCE_GEN=1 timeout 120 tools/bin/attn_bench 7 144 32 0 3 1 1 1 2 2>&1 | grep -E "check" >> $out
done
for rep in 1 2; do for sm in 0 1 2; do for G in 256 512; do echo -n "SMALL=$sm " >> $out; DA_ATTN_RES_SMALL=$sm timeout 120 tools/bin/attn_bench $G 144 32 0 50 0 0 1 2 2>&1 | grep "G=" >> $out; done; done; done
for rep in 1 2; do for sm in 0 1 2; do echo -n "config 2 SMALL=$sm " >> $out; DA_ATTN_RES_SMALL=$sm python bench.py --config 2 --no-cpu-baseline --no-parity-mode --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out; done; done
cat $out
