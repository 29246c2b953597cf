#!/bin/bash
# GPU box: SQ counters of the attention kernels of the headline step through tools/bin/attn_bench (standalone launches of
# da::launch_attn_dense on the benched shapes, production dispatch): MFMA busy cycles, VALU / MFMA instruction counts, waits.
# Counters in their own passes with --kernel-trace only (MI355X_MICROARCH.md).   usage: bash tools/collect_attn_pmc.sh
set -u
ROUND=${ROUND:-r06}
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; cd /tmp
F=$OUT/${ROUND}_pmc_attention_sq.txt; : > $F
for GP in 64 32; do       # 64 puzzles per launch (one-branch loop) and 32 (what each branch of the default two-branch graph launches)
for CASE in "32 0 2" "144 1 2" "144 1 1"; do        # C, fold, kernel choice (2 = production dispatch: the optimistic kernels k_attn_optt; 1 = k_attn_dual)
  set -- $CASE
  if [ "$GP" = 32 ] && [ "$3" = 1 ]; then continue; fi
  for GRP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    W=/tmp/pmc_attn_${GP}_$1_$3_$(echo $GRP | cut -c1-12 | tr ' ' _); rm -rf $W; mkdir -p $W
    timeout 600 rocprofv3 --kernel-trace --pmc $GRP -d $W -o p -- $REPO/tools/bin/attn_bench $GP 900 $1 $2 5 0 0 1 $3 > $W/log 2>&1
    DB=$(find $W -name "*results.db" | head -1)
    echo "== G=$GP C=$1 fold=$2 kernel=$3 : $GRP" >> $F
    python $REPO/profiles/rocpd_pmc.py $DB k_attn >> $F 2>&1
  done
done
done
tail -40 $F
