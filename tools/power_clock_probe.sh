#!/bin/bash
# GPU box: package power (rocm-smi) AND the effective shader clock (tools/bin/clock_probe, a one-wave second process) while one
# kernel class -- or the whole captured step -- runs back to back.  Answers "where do the step's joules go" and "how far is each
# class clock-throttled" (the step sits at the 1400 W package cap).   usage: bash tools/power_clock_probe.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
sample() { sleep 5; tools/bin/clock_probe 3 500 & cp=$!; for i in 1 2 3; do rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | head -1 | tr '\n' ' '; sleep 1; done; wait $cp; }
run() { tag=$1; shift; ( "$@" > /tmp/kp.log 2>&1 ) & pid=$!; echo "== $tag"; sample; wait $pid; tail -1 /tmp/kp.log | cut -c1-170; }
echo "idle:"; rocm-smi --showpower --showmaxpower 2>/dev/null | grep -i "power" | head -3; tools/bin/clock_probe 1 500
run "attn hidden (k_attn_optt<32>), 32 puzzles"          tools/bin/attn_bench 32 900 32 0 60000 0 0 1 2
run "attn hidden, 64 puzzles"                            tools/bin/attn_bench 64 900 32 0 32000 0 0 1 2
run "attn last (k_attn_optt<144,fold>), 32 puzzles"      tools/bin/attn_bench 32 900 144 1 36000 0 0 1 2
run "attn last, 64 puzzles"                              tools/bin/attn_bench 64 900 144 1 18000 0 0 1 2
DA_OPT_HID=10 run "attn hidden PIPELINED, 32 puzzles"    tools/bin/attn_bench 32 900 32 0 60000 0 0 1 2
DA_OPT_LAST=10 run "attn last PIPELINED, 32 puzzles"     tools/bin/attn_bench 32 900 144 1 36000 0 0 1 2
run "projection conv 0: 28800 x 1152 -> 1024"            python tools/gemm_probe_loop.py 28800 1152 1024 90000
run "projection conv 1/2: 28800 x 256 -> 1024"           python tools/gemm_probe_loop.py 28800 256 1024 150000
run "projection conv 3: 28800 x 256 -> 2560"             python tools/gemm_probe_loop.py 28800 256 2560 90000
run "whole step, two-branch graph (default)"             python bench.py --steps 100 --warmup 10 --no-parity-mode --no-cpu-baseline --no-roofline --replays 150
DA_TWO_BRANCH=0 run "whole step, one-branch graph"       python bench.py --steps 100 --warmup 10 --no-parity-mode --no-cpu-baseline --no-roofline --replays 150
