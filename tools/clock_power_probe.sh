#!/bin/bash
# sample the shader clock and power while the attention layer runs in a loop
( G=64 CS=32 ITERS=250000 timeout 90 python tools/attn_probe.py > /tmp/loop.log 2>&1 ) &
sleep 20
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4; sleep 1; done
wait
tail -1 /tmp/loop.log
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
