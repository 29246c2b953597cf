#!/bin/bash
# GPU box: package power (rocm-smi, 1 Hz) while ONE kernel runs back to back -- which kernels of the step are the expensive ones
# in energy (power x time per launch), now that the step as a whole sits at the 1400 W cap.
sample() { sleep 4; for i in 1 2 3 4; do rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | head -1 | tr '\n' ' '; sleep 1; done; echo; }
run() { tag=$1; shift; ( "$@" > /tmp/kp.log 2>&1 ) & pid=$!; echo -n "$tag: "; sample; wait $pid; tail -1 /tmp/kp.log | cut -c1-150; }
run "attn_opt (hidden, C=32)"        tools/bin/attn_bench 64 900 32 0 14000 0 0 1 2
run "attn_dense FAST (last, C=144)"  tools/bin/attn_bench 64 900 144 1 7000 0 0 1 0
run "attn_dual (last, C=144)"        tools/bin/attn_bench 64 900 144 1 7000 0 0 1 2
run "gemm conv3 57600x256x2560"      python tools/gemm_probe_loop.py 57600 256 2560 60000
run "gemm hidden 57600x256x1024"     python tools/gemm_probe_loop.py 57600 256 1024 150000
