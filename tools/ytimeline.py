#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the last YOLO forward in a rocprofv3 --kernel-trace database (rocpd SQLite).
usage: ytimeline.py results.db [first-kernel-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x, lds_size from kernels order by start").fetchall()
i0 = [i for i, r in enumerate(rows) if "stem_conv" in r[0] or "front_fused" in r[0]][-1]
if "front_fused" in rows[i0][0]:
    i0 += 1                               # (the fused front end has no separate preprocess launch before it)
t0 = rows[i0][1]
for j, r in enumerate(rows[i0 - 1:]):
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)E", nm)
    nm = m.group(1) if m else nm
    print("%2d %-40s %7.1f us  start %8.1f  end %8.1f  wgs %6d x%d  lds %6d" % (
        j, nm[:40], (r[2] - r[1]) / 1e3, (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, r[3] // max(r[5], 1), r[4], r[6]))
    if "nms_greedy" in r[0]:
        break
