#!/bin/bash
# k_attn_res experiment switches (DA_ATTN_RES_PH): 1 default, 5 two PV accumulators, 6 younger waves at priority 1, 7 both
cd "$(dirname "$0")/.."
out=gpurun_out/r05_res_probe4.log
: > $out
for ph in 1 5 6 7; do
  echo "== PH=$ph" >> $out
  DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 1 1 2 2>&1 | grep -E "check" >> $out
  DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 40 2 2>&1 | grep -E "check" >> $out
  for rep in 1 2; do for G in 32 64; do echo -n "PH=$ph " >> $out; DA_ATTN_RES_PH=$ph timeout 120 tools/bin/attn_bench $G 900 32 0 50 0 0 1 2 2>&1 | grep "G=" >> $out; done; done
done
cat $out
