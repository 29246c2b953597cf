set -x
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "bf16_mma or backward_matches_oracle or real_ddp or data_parallel_training_two" 2>&1 | tail -5
for rep in 1 2; do
for cfg in "DA_GGEMM_SMALL=1" "DA_GGEMM_SMALL=0"; do
  timeout 300 env $cfg python bench.py --config 5 --precision bf16 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 bf16 $cfg', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'])"
done
done
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --precision bf16 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
head -24 gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt | cut -c1-150
