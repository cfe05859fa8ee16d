#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max ns, share) from a rocprofv3 rocpd SQLite file, written as the
same CSV `rocprofv3 --stats --output-format csv` would give.  usage: rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                  "from kernels group by name order by sum(end - start) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for n, c, t, a, mn, mx in rows:
    out.writerow([n, c, int(t), round(a, 1), round(100.0 * t / tot, 2), int(mn), int(mx)])
