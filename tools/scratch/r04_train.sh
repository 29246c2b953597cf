set -x
python -m pytest tests/test_gpu_train.py -x -q -k "bf16_mma or backward_matches_oracle or p_losses_gradients" 2>&1 | tail -8
python -m pytest tests/test_gpu_softmax_fallbacks.py -q -k "offset_invariant" 2>&1 | tail -4
for rep in 1 2; do
for prec in fp32 bf16; do
  python bench.py --config 5 --precision $prec --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('config5 $prec', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'], round(d['roofline']['frac'],4))"
done
done
for prec in fp32 bf16; do
  python bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision $prec --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print('exo900 $prec', round(d['value']), round(d['ms_per_step'],3), d['phases_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --precision bf16 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1
python $GRAFT_REPO_ROOT/profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/r04_rocprof_kernel_stats_config5_bf16mma.txt
