#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
F=$OUT/r05_pmc_pcd_knn.txt; : > $F
for GRP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
  W=/tmp/pmc_pcd_$(echo $GRP | cut -c1-12 | tr ' ' _); rm -rf $W; mkdir -p $W
  DA_PCD_TWO_STREAMS=0 DA_PCD_KNN_QB=32 timeout -k 5 400 rocprofv3 --kernel-trace --pmc $GRP -d $W -o p -- python $REPO/bench.py --mode encode --config 4 --no-cpu-baseline --steps 3 --warmup 1 > $W/log 2>&1
  DB=$(find $W -name "*results.db" | head -1)
  echo "== $GRP" >> $F
  python $REPO/profiles/rocpd_pmc.py $DB k_pcd >> $F 2>&1
done
