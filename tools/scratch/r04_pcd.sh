set -x
timeout 900 python -m pytest tests/test_gpu_pcd_encoder.py -x -q 2>&1 | tail -4
for rep in 1 2; do
timeout 300 python bench.py --mode encode --config 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
python -c "import json; d=json.load(open('/tmp/o.json')); print('pcd_encode', round(d['value']), round(d['ms_per_step'],3))"
done
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pcd -o s -- python $GRAFT_REPO_ROOT/bench.py --mode encode --config 4 --no-cpu-baseline > /tmp/prof_pcd.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_pcd -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_pcd_encode.txt 2>&1
head -12 gpurun_out/r04_rocprof_kernel_stats_pcd_encode.txt | cut -c1-150
