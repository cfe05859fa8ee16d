#!/usr/bin/env python3
"""Launch-by-launch timeline of the last float32-mode forward in a rocprofv3 --kernel-trace database: usage ytimeline32.py results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
i0 = [i for i, r in enumerate(rows) if "preprocess_f32" in r[0]][-1]
tot = 0.0
for j, r in enumerate(rows[i0:]):
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    d = (r[2] - r[1]) / 1e3
    tot += d
    print("%2d %-36s %8.1f us  wgs %6d x%d" % (j, nm[:36], d, r[3] // max(r[5], 1), r[4]))
    if "nms_greedy" in r[0]:
        break
print("sum %.1f us" % tot)
