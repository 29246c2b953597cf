#!/usr/bin/env python
"""Print the per-position average duration of the repeating kernel sequence of one denoising step
from a rocprofv3 rocpd database.  usage: python profiles/rocpd_sequence.py <results.db> <kernels_per_step>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2])
rows = con.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
rows = [r for r in rows if "da::" in r[0]]            # the library's kernels only (torch helpers come and go)
names = [r[0] for r in rows]
# the steady state is the tail: take the last 10 full steps
tail = rows[-per * 10:]
print(f"{'#':>3s} {'kernel':70s} {'grid':>12s} {'avg_us':>8s} {'gap_us':>7s}")
tot = 0.0
for i in range(per):
    seg = tail[i::per]
    assert len({s[0] for s in seg}) == 1, "sequence does not repeat with this period"
    avg = sum(s[2] - s[1] for s in seg) / len(seg) / 1e3
    idx = [len(rows) - per * 10 + i + per * k for k in range(10)]
    gap = sum(rows[j][1] - rows[j - 1][2] for j in idx) / len(idx) / 1e3
    tot += avg
    nm = seg[0][0].replace("void ", "").replace("da::", "")[:70]
    print(f"{i:3d} {nm:70s} {str(seg[0][3]) + 'x' + str(seg[0][4]):>12s} {avg:8.1f} {gap:7.1f}")
print(f"sum of kernel time per step: {tot:.1f} us")
