set -u
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for MODE in 2 13 12; do
  for GRP in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
    W=/tmp/pmc_st_${MODE}_$(echo $GRP | cut -c1-10 | tr ' ' _); rm -rf $W; mkdir -p $W
    rocprofv3 --kernel-trace --pmc $GRP -d $W -o p -- $REPO/tools/bin/store_probe 57600 2560 1 512 $MODE > $W/log 2>&1
    DB=$(find $W -name "*results.db" | head -1)
    echo "== mode $MODE : $GRP"
    python $REPO/profiles/rocpd_pmc.py $DB k_store 2>&1 | awk '/^k_store/{n=split($0,a," +"); print "   ", a[n-5], a[n-3], a[n]}'
  done
done
