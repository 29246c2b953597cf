#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 2400 python -m pytest tests -q -m gpu > $O/r05_final_tests.log 2>&1
echo "tests rc=$?" >> $O/r05_final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_final_smoke.log 2>&1
echo "smoke rc=$?" >> $O/r05_final_smoke.log
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/r05_bench_$tag.json 2> $O/r05_bench_$tag.err; echo "bench $tag rc=$?"; }
b config_5 --config 5
b config_5_fp32 --config 5 --precision fp32
b config_5_exophormer_d539 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16
b config_5_exophormer_d539_fp32 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision fp32
b scripted --config scripted
( cd /tmp && rm -rf /tmp/prof_t5 && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $O/r05_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
timeout 120 python tools/step_timeline.py $(find /tmp/prof_t5 -name "*results.db" | head -1) k_af_a 5 10 > $O/r05_config5_step_timeline_final.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_exo && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) > $O/r05_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt 2>&1
