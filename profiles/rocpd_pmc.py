#!/usr/bin/env python
"""Average PMC counter values per kernel from a rocprofv3 rocpd database collected with
``rocprofv3 --kernel-trace --pmc <COUNTERS> ...`` (one counter group per run).
usage: python profiles/rocpd_pmc.py <results.db> [kernel-name substring]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = con.execute(
    "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
    "from counters_collection where kernel_name like ? group by kernel_name, counter_name "
    "order by kernel_name, counter_name", (f"%{sub}%",)).fetchall()
print(f"{'kernel':56s} {'counter':28s} {'n':>5s} {'avg':>16s} {'min':>16s} {'max':>16s} {'avg_us':>9s}")
for r in rows:
    name = r[0].replace("void ", "").replace("da::", "")[:56]
    print(f"{name:56s} {r[1]:28s} {r[2]:5d} {r[3]:16.1f} {r[4]:16.1f} {r[5]:16.1f} {r[6] / 1e3:9.1f}")
