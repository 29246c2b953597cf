#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 1200 python -m pytest tests/test_gpu_pcd_encoder.py -x -q -m gpu > gpurun_out/r05_p3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_p3_tests.log
{
for rep in 1 2; do
for cfg in "16 1" "32 1" "16 0"; do set -- $cfg
  DA_PCD_KNN_QB=$1 DA_PCD_TWO_STREAMS=$2 timeout -k 5 300 python bench.py --mode encode --config 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('pcd encode qb=$1 two_streams=$2', round(d['value']), d['unit'], round(d['ms_per_step'],3), 'ms')"
done
done
} > gpurun_out/r05_p3_ab.log 2>&1
( cd /tmp && rm -rf /tmp/prof_p && DA_PCD_TWO_STREAMS=0 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o s -- python $GRAFT_REPO_ROOT/bench.py --mode encode --config 4 --no-cpu-baseline > /tmp/prof_p.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_p -name "*results.db" | head -1) 2>&1 | head -12 > gpurun_out/r05_p3_stats.txt
