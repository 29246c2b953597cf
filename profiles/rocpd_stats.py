#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) per kernel and per launch shape.
usage: python profiles/rocpd_stats.py <results.db> [min_count]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute(
    "select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
    "from kernels group by name, grid_x, grid_y order by sum(end-start) desc").fetchall()
tot = sum(r[8] for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[4] for r in rows)} dispatches")
print(f"{'kernel':60s} {'grid':>14s} {'n':>5s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
for r in rows[:40]:
    name = r[0].replace("void ", "").replace("da::", "")[:60]
    print(f"{name:60s} {str(r[1]) + 'x' + str(r[2]):>14s} {r[4]:5d} {r[5] / 1e3:9.1f} {r[6] / 1e3:9.1f} {r[7] / 1e3:9.1f} {100 * r[8] / tot:5.1f}%")
