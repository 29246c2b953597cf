#!/bin/bash
# GPU box, round 6 call 1: co-residency probe (thin projection beside the resident attention), thin projections in the model
# (parity subset + headline A/B at the driver's arguments), concurrent timeline of the probe, step-rule power probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 tools/bin/corun_probe 32 900 200 > $O/r06_corun_probe.log 2>&1; echo "corun rc=$?"; cat $O/r06_corun_probe.log
# parity of the model with thin projections: 900-piece forward / trajectories vs the reference fixtures, two-branch identities
DA_GEMM_THIN=1 timeout 900 python -m pytest tests/test_gpu_benched_mode.py -m gpu -x -q -k "900 or rot900 or two_branch or pair" > $O/r06_thin_parity.log 2>&1; echo "thin parity rc=$?"; tail -3 $O/r06_thin_parity.log
L=$O/r06_thin_ab_steps20.log; : > $L
run() { echo "$1 $(env $1 timeout 120 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'], d['value'])")" >> $L; }
for i in 1 2 3 4 5; do run DA_GEMM_THIN=0 20; run DA_GEMM_THIN=1 20; done
echo "--- steps 100" >> $L
for i in 1 2 3; do run DA_GEMM_THIN=0 100; run DA_GEMM_THIN=1 100; done
cat $L
# kernel timeline of the probe (start / end per dispatch: which kernels overlap)
( cd /tmp && rm -rf /tmp/prof_corun && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_corun -o s -- $GRAFT_REPO_ROOT/tools/bin/corun_probe 32 900 30 > /tmp/prof_corun.log 2>&1 )
python profiles/rocpd_stats.py $(find /tmp/prof_corun -name "*results.db" | head -1) > $O/r06_rocprof_kernel_stats_corun_probe.txt 2>&1
cp $(find /tmp/prof_corun -name "*results.db" | head -1) $O/r06_corun_probe_results.db 2>/dev/null
ROUND=r06 timeout 600 bash tools/step_rule_power_probe.sh > /dev/null 2>&1
tail -40 $O/r06_step_rule_power.log
