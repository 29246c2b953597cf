#!/bin/bash
# GPU box: everything the round's tables quote, in one call -- rocprofv3 kernel statistics + PMC traffic of the headline and the
# configuration-3 variants, SQ counters of the attention and projection kernels, and one bench line per configuration / mode.
# Outputs go to gpurun_out/<round>_*; copy them to profiles/<round>/ afterwards.    usage: ROUND=r03 bash tools/collect_round.sh
set -u
export ROUND=${ROUND:-r06}
O=gpurun_out
bash tools/collect_profiles.sh headline --config 3p
bash tools/collect_profiles.sh headline_half --config 3p --puzzles 32     # the launch shape of each branch of the default two-branch loop
DA_TWO_BRANCH=1 bash tools/collect_profiles.sh headline_two_branch --config 3p
bash tools/collect_profiles.sh config3_d539 --config 3
bash tools/collect_profiles.sh config3_d90 --config 3 --degree 90
DA_HYBRID=off bash tools/collect_profiles.sh config3_d539_csr_only --config 3 --steps 4
DA_HYBRID=off bash tools/collect_profiles.sh config3_d90_csr_only --config 3 --degree 90 --steps 4
bash tools/collect_profiles.sh scripted --config scripted --no-train-side
bash tools/collect_profiles.sh csr --config csr
bash tools/collect_attn_pmc.sh > /dev/null 2>&1
# FETCH_SIZE calibration for the edge-list kernels' row gathers (known byte counts; tools/gather_probe.hip)
for M in 0 1; do
  W=/tmp/pmc_gather_$M; rm -rf $W; mkdir -p $W
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $W -o p -- $GRAFT_REPO_ROOT/tools/bin/gather_probe $M 1024 256 > $W/log 2>&1 )
  { tail -1 $W/log; python profiles/rocpd_pmc.py $(find $W -name "*results.db" | head -1) k_gather; } >> $O/${ROUND}_pmc_gather_calibration.txt 2>&1
done
# the counter tables feed the bench lines' `traffic` / `mfma_busy_counter` fields: summarise them BEFORE the lines are produced
mkdir -p profiles/$ROUND
cp $O/${ROUND}_pmc_*.txt $O/${ROUND}_rocprof_*.txt profiles/$ROUND/ 2>/dev/null
python tools/make_pmc_json.py $ROUND > $O/${ROUND}_make_pmc_json.log 2>&1
cp profiles/$ROUND/pmc_traffic.json profiles/$ROUND/pmc_attention_sq.json $O/ 2>/dev/null
b() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/${ROUND}_bench_$tag.json 2> $O/${ROUND}_bench_$tag.err; tail -c 300 $O/${ROUND}_bench_$tag.json | head -c 0; echo "bench $tag rc=$?"; }
b config_3p --steps 100 --warmup 10
b config_1 --config 1
b config_2 --config 2
b config_3 --config 3
b config_3_d90 --config 3 --degree 90
b config_4 --config 4
b scripted --config scripted
b csr --config csr
# training lines: the benched default is the bf16-operand mode (--precision fp32 = the reference's arithmetic)
b config_5 --config 5
b config_5_fp32 --config 5 --precision fp32
b config_5_exophormer_d539 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16
b config_5_exophormer_d539_fp32 --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --precision fp32
b config_5_pixels --config 5 --pixels
b config_5_pixels_fp32 --config 5 --pixels --precision fp32
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_t5 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t5 -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /tmp/prof_t5.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_t5 -name "*results.db" | head -1) > $O/${ROUND}_rocprof_kernel_stats_config5_bf16mma.txt 2>&1
( cd /tmp && rm -rf /tmp/prof_exo && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_exo -o s -- python $GRAFT_REPO_ROOT/bench.py --config 5 --arch exophormer --train-side 30 --degree 539 --train-puzzles 16 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_exo.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_exo -name "*results.db" | head -1) > $O/${ROUND}_rocprof_kernel_stats_config5_exophormer_d539_bf16mma.txt 2>&1
b e2e --mode e2e
b encode --mode encode
b pcd_encode --mode encode --config 4
( cd /tmp && rm -rf /tmp/prof_pcd && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_pcd -o s -- python $GRAFT_REPO_ROOT/bench.py --mode encode --config 4 --no-cpu-baseline > /tmp/prof_pcd.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_pcd -name "*results.db" | head -1) > $O/${ROUND}_rocprof_kernel_stats_pcd_encode.txt 2>&1
b config_3p_cpu_1_thread --steps 20 --warmup 5 --cpu-baseline-full --no-parity-mode --replays 5
ls $O | grep ${ROUND}_ | wc -l
