#!/bin/bash
# GPU box: SQ counters of k_conv_fused (and the two-kernel path next to it) over tools/attn_probe.py
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; cd /tmp
for MODE in fused unfused; do
  W=/tmp/pmc_$MODE; rm -rf $W; mkdir -p $W
  if [ $MODE = fused ]; then export DA_CONV_FUSED=1; else unset DA_CONV_FUSED; fi
  CS=32 ITERS=5 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $W -o p -- python $REPO/tools/attn_probe.py > $W/log 2>&1
  DB=$(find $W -name "*results.db" | head -1)
  echo "== $MODE" >> $OUT/r2_pmc_fused.txt
  python $REPO/profiles/rocpd_pmc.py $DB k_ >> $OUT/r2_pmc_fused.txt 2>&1
  W2=/tmp/pmc2_$MODE; rm -rf $W2; mkdir -p $W2
  CS=32 ITERS=5 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT -d $W2 -o p -- python $REPO/tools/attn_probe.py > $W2/log 2>&1
  DB=$(find $W2 -name "*results.db" | head -1)
  python $REPO/profiles/rocpd_pmc.py $DB k_ >> $OUT/r2_pmc_fused.txt 2>&1
  tail -3 $W2/log >> $OUT/r2_pmc_fused.txt
done
cat $OUT/r2_pmc_fused.txt
