#!/usr/bin/env python3
"""Which hardware queue every stream of a traced run was served by (rocprofv3 --kernel-trace database): usage qmap.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
seen = {}
for qid, sid, n, c in db.execute("select queue_id, stream_id, name, count(*) from kernels group by queue_id, stream_id, name having count(*) > 50 order by queue_id, stream_id"):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:26]
    seen.setdefault((qid, sid), []).append(n)
for k, v in seen.items():
    print("  queue", k[0], "stream", k[1], ":", ", ".join(v[:5]), "... (%d kernels)" % len(v))
