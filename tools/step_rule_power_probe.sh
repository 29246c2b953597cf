#!/bin/bash
# Runs ON THE GPU BOX (gpurun), ~4 min: shader clock and package power of the headline loop under the four combinations of the large-graph step
# rule (DESIGN § "Measured, round 5" (5): row-panel projections x next-step embedding in the tail kernel), at both loop lengths -- the question
# the round-5 A/Bs left open is WHY the pair gains 3 % on 100-iteration loops and nothing on 20-iteration ones (energy at the package cap?).
#   usage: bash tools/step_rule_power_probe.sh            -> gpurun_out/<ROUND>_step_rule_power.log
set -u
ROUND=${ROUND:-r06}
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
L=gpurun_out/${ROUND}_step_rule_power.log; : > $L
sample() { for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -3 | tr '\n' ' '; echo; sleep 1; done; }
for steps in 100 20; do
  for combo in "DA_ENABLE_XPANEL=0 DA_TAIL_NEXT=0" "DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=0" "DA_ENABLE_XPANEL=0 DA_TAIL_NEXT=1" "DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=1"; do
    echo "== steps $steps | $combo" >> $L
    # enough back-to-back passes for ~20 s of steady state (a pass is steps x ~0.68 ms)
    reps=$(( 20000 / (steps * 68 / 100 + 1) ))
    ( env $combo timeout 120 python bench.py --steps $steps --warmup 5 --no-parity-mode --no-cpu-baseline --no-roofline --replays $reps > /tmp/loop.log 2>&1 ) &
    sleep 12
    sample >> $L
    wait
    python -c "import json; d=json.loads([l for l in open('/tmp/loop.log') if l.startswith('{')][-1]); print('median ms_per_step', d['ms_per_step'], 'value', d['value'])" >> $L 2>&1
  done
done
echo idle: >> $L; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|power" | head -5 >> $L
cat $L
