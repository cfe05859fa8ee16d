#!/usr/bin/env python3
"""Per-kernel counter sums from a rocprofv3 --pmc run (rocpd SQLite): for every kernel name, launches and, per counter,
the mean value per launch.  usage: rocpd_pmc.py results.db [name-substring]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
out = {}
for k, c, n, v, d in rows:
    if pat and pat not in k:
        continue
    e = out.setdefault(k.split("(")[0][-60:], {"launches": n, "avg_ns": round(d or 0, 1)})
    e[c] = v
print(json.dumps(out, indent=1))
