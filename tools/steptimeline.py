#!/usr/bin/env python3
"""Launch-by-launch timeline of one config-3 step (synth_rows_kernel to the next synth_rows_kernel) of a rocprofv3 --kernel-trace
database, every queue: shows which lane / detector-tail kernels really run beside which convolutions.
usage: steptimeline.py results.db [steps-back-from-the-last, default 12]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = db.execute("select name,start,end,grid_x,grid_y,workgroup_x,lds_size,queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "synth" in r[0]]
a, b = idx[-back], idx[-back + 1]
t0, t1 = rows[a][1], rows[b][1]


def nm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)E", n)
    return (m.group(1) if m else n)[:44]


for r in rows[a:]:
    if r[1] >= t1:
        break
    print("%-44s q%-3s %7.1f us  start %8.1f  end %8.1f  wgs %6d  lds %6d" % (
        nm(r[0]), r[7], (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[3] // max(r[5], 1)) * r[4], r[6]))
print("step %.1f us" % ((t1 - t0) / 1e3))
