set -x
export TMPDIR=/tmp
for deg in 539 90; do
W=/tmp/prof_c3_$deg; rm -rf $W; mkdir -p $W
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $W -o s -- python $GRAFT_REPO_ROOT/bench.py --config 3 --degree $deg --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --replays 0 > $W/log 2>&1 )
python profiles/rocpd_stats.py $(find $W -name "*results.db" | head -1) > gpurun_out/r04c_rocprof_kernel_stats_config3_d$deg.txt 2>&1
head -14 gpurun_out/r04c_rocprof_kernel_stats_config3_d$deg.txt | cut -c1-150
done
