#!/usr/bin/env python
"""Per-step timeline of a rocprofv3 kernel trace (rocpd ``*_results.db``): steps are cut at every launch of a marker kernel
(default k_af_a: the first kernel of FusedAdafactor.step, once per optimizer step); prints, for the steady-state steps, the
span from marker to marker, the sum of kernel durations inside it, the idle time between kernels and the kernels of one step
in launch order.   usage: python tools/step_timeline.py <results.db> [marker] [first_step] [n_steps]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_af_a"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 5
nst = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rows = con.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
cuts = [i for i, r in enumerate(rows) if marker in r[0]]
print(f"{len(rows)} dispatches, {len(cuts)} marker launches")
spans, busys = [], []
for s in range(first, min(first + nst, len(cuts) - 1)):
    a, b = cuts[s], cuts[s + 1]
    span = rows[b][1] - rows[a][1]
    busy = sum(r[2] - r[1] for r in rows[a:b])
    spans.append(span / 1e3)
    busys.append(busy / 1e3)
print("span us   :", " ".join(f"{x:7.1f}" for x in spans))
print("kernels us:", " ".join(f"{x:7.1f}" for x in busys))
if spans:
    print(f"mean span {sum(spans) / len(spans):.1f} us, kernels {sum(busys) / len(busys):.1f} us, idle {100 * (1 - sum(busys) / sum(spans)):.1f} %")
    a, b = cuts[first], cuts[first + 1]
    t0 = rows[a][1]
    prev_end = None
    print(f"{'start':>8s} {'dur':>7s} {'gap':>6s}  kernel (step {first})")
    for r in rows[a:b]:
        gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} {gap:6.1f}  {r[0].replace('void ', '').replace('da::', '')[:90]} [{r[3]}x{r[4]}]")
        prev_end = r[2]
