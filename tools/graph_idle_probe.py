#!/usr/bin/env python
"""GPU busy / idle time inside the captured sampling loop, from a rocprofv3 kernel trace (rocpd database) of
`python bench.py --steps 100 --warmup 10 --replays 2 --no-parity-mode --no-cpu-baseline --no-roofline`:
the union of the kernels' [start, end) intervals over the LAST graph replay against its wall span, and the time during
which kernels of both branches overlap.   usage: python tools/graph_idle_probe.py <results.db>
Finding (round 3): under --kernel-trace the dispatches are serialised -- >= 2 kernels in flight for 1 % of the span, idle 0.7 %,
and the two-branch loop then measures SLOWER than the one-branch loop (0.833 vs 0.768 ms per step) because its half-size kernels
run one after the other; the +3.4 % of the two-branch loop exists only untraced (HIP events around whole replays)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
rows = [r for r in rows if "da::" in r[0]]
# the last replay = the last 100 steps: find them by counting tail kernels (one per step per branch) backwards
tail_idx = [i for i, r in enumerate(rows) if "k_tail_fused" in r[0]]
n_tail = 200 if len(tail_idx) >= 400 else 100
first = tail_idx[-n_tail]
# start at the first embed kernel at or before the first of those tails
seg = rows[max(0, first - 40):]
emb = [i for i, r in enumerate(seg) if "k_embed_pos_time" in r[0]]
seg = seg[emb[0]:]
t0, t1 = seg[0][1], max(r[2] for r in seg)
ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
busy = over = 0
depth = 0
last = t0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d
    last = t
span = t1 - t0
print(f"kernels {len(seg)}, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us ({busy / span:.3f}), idle {(span - busy) / 1e3:.1f} us, "
      f">= 2 kernels in flight {over / 1e3:.1f} us ({over / span:.3f}); sum of kernel durations {sum(r[2] - r[1] for r in seg) / 1e3:.1f} us")
