#!/bin/bash
# k_attn_res: uniform-slot fast path (DMA issued without waiting for the graph table) -- parity on uniform and ragged Batches, stamps, timing
cd "$(dirname "$0")/.."
out=gpurun_out/r05_res_probe5.log
: > $out
for n in 900 897 513 1216; do for nd in 0 1; do timeout 120 tools/bin/attn_bench 3 $n 32 0 3 1 $nd 1 2 2>&1 | grep -E "check" >> $out; done; done
timeout 120 tools/bin/attn_bench 3 900 32 0 3 1 0 40 2 2>&1 | grep -E "check" >> $out
PROBE3=1 timeout 120 tools/bin/attn_bench_probe 32 900 32 0 20 0 0 1 2 2>&1 | grep -E "probe3" >> $out
for rep in 1 2 3; do for G in 32 64; do timeout 120 tools/bin/attn_bench $G 900 32 0 50 0 0 1 2 2>&1 | grep "G=" >> $out; done; done
python -m pytest tests/test_gpu_attn_resident.py -x -q -m gpu 2>&1 | tail -2 >> $out
cat $out
