#!/bin/bash
# GPU box: SQ counters of the projection kernels (k_gemm_wreg / k_gemm_wreg2) through tools/gemm_probe.py, one shape per pass;
# counters in their own passes with --kernel-trace only.   usage: bash tools/collect_gemm_pmc.sh [tag]   (env: DA_WREG_DIRECT, DA_GEMM_DEBUG ...)
set -u
ROUND=${ROUND:-r03}
TAG=${1:-default}
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; cd /tmp
F=$OUT/${ROUND}_pmc_gemm_sq_$TAG.txt; : > $F
for CASE in "57600 256 2560" "57600 256 1024"; do
  set -- $CASE
  for GRP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    W=/tmp/pmc_gemm_$3_$(echo $GRP | cut -c1-12 | tr ' ' _); rm -rf $W; mkdir -p $W
    rocprofv3 --kernel-trace --pmc $GRP -d $W -o p -- python $REPO/tools/gemm_probe.py $1 $2 $3 > $W/log 2>&1
    DB=$(find $W -name "*results.db" | head -1)
    echo "== M=$1 K=$2 N=$3 : $GRP" >> $F
    python $REPO/profiles/rocpd_pmc.py $DB k_gemm_ >> $F 2>&1
  done
done
cat $F | cut -c1-170
