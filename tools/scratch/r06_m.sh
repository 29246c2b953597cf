#!/bin/bash
# GPU box: FETCH_SIZE calibration on random row gathers (the edge-list kernel's pattern) and the csr / configuration-3 lines in the final JSON form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
: > $O/r06_pmc_gather_calibration.txt
for M in 0 1; do
  W=/tmp/pmc_gather_$M; rm -rf $W; mkdir -p $W
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W -o p -- $GRAFT_REPO_ROOT/tools/bin/gather_probe $M 1024 256 > $W/log 2>&1 )
  { tail -1 $W/log; python profiles/rocpd_pmc.py $(find $W -name "*results.db" | head -1) k_gather; } >> $O/r06_pmc_gather_calibration.txt 2>&1
done
cat $O/r06_pmc_gather_calibration.txt
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err; echo "bench $tag rc=$?"; }
b csr --config csr
b config_3 --config 3
b config_3_d90 --config 3 --degree 90
python - <<'PY'
import json
for t in ("csr","config_3","config_3_d90"):
    d=json.loads([l for l in open(f"gpurun_out/r06_bench_{t}.json") if l.startswith("{")][-1])
    r=d["roofline"]; print(t, round(d["ms_per_step"],4), r["bound"], r["kernel"], round(r["achieved"],1), r["unit"], "frac", round(r["frac"],4), "traffic", r.get("traffic"), "mem-side", r.get("memory_side_GBps"))
PY
