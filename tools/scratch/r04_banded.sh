# (every command under its own timeout: a crashing application under rocprofv3 hung a whole 40-minute call earlier in the round)
set -x
timeout 600 python -m pytest tests/test_gpu_softmax_fallbacks.py -x -q -k "hybrid" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_benched_mode.py -x -q -k "exo900 or banded or expander_mask or config3" 2>&1 | tail -8
for rep in 1 2; do
  for cfg in "DA_EXPANDER_LAYOUT=banded" "DA_EXPANDER_LAYOUT=natural" "DA_EXPANDER_LAYOUT=natural DA_ATTN_OPT_MASKED=0"; do
    for deg in 539 90; do
    timeout 300 env $cfg python bench.py --config 3 --degree $deg --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > /tmp/o.json
    python -c "import json; d=json.load(open('/tmp/o.json')); print('$cfg d=$deg', round(d['value']), round(d['ms_per_step'],4))"
    done
  done
done
set -x
export TMPDIR=/tmp
for deg in 539 90; do
W=/tmp/prof_c3_$deg; rm -rf $W; mkdir -p $W
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $W -o s -- python $GRAFT_REPO_ROOT/bench.py --config 3 --degree $deg --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --no-roofline --replays 0 > $W/log 2>&1 )
python profiles/rocpd_stats.py $(find $W -name "*results.db" | head -1) > gpurun_out/r04c_rocprof_kernel_stats_config3_d$deg.txt 2>&1
head -14 gpurun_out/r04c_rocprof_kernel_stats_config3_d$deg.txt | cut -c1-150
done
