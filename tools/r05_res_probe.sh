#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r05_res_probe3.log
: > $out
for ph in 1 3 4; do
  echo "== PH=$ph (1: queue + K prefetch, 3: fixed slabs + K prefetch, 4: queue, no K prefetch)" >> $out
  for n in 900 897 1216 513; do for nd in 0 1; do DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench 3 $n 32 0 3 1 $nd 1 2 2>&1 | grep -E "check" >> $out; done; done
  DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 40 2 2>&1 | grep -E "check" >> $out
  DA_ATTN_RES_PH=$ph DA_ATTN_FORCE_GEN=1 timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 1 1 2 2>&1 | grep -E "check" >> $out
  DA_ATTN_RES_PH=$ph PROBE3=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 2>&1 | grep -E "probe3" >> $out
  for G in 5 32 64; do echo -n "PH=$ph " >> $out; DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench $G 900 32 0 50 0 0 1 2 2>&1 | grep "G=" >> $out; done
done
for G in 32 64; do echo -n "ring " >> $out; DA_ATTN_RES=0 timeout 120 tools/bin/attn_bench $G 900 32 0 50 0 0 1 2 2>&1 | grep "G=" >> $out; done
cat $out
