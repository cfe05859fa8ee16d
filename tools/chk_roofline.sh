#!/bin/bash
# bench.py's event-timed launch duration of the headline kernel beside rocprofv3's kernel duration for the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/chk
rocprofv3 --kernel-trace --stats -d gpurun_out/chk -o c -- python3 bench.py --no-also --no-cpu-baseline "$@" > gpurun_out/chk.json 2> gpurun_out/chk.err
python3 - <<'PY'
import json, glob, csv
j = json.load(open("gpurun_out/chk.json"))
print("bench:", j["value"], "frames/s,", j["ms_per_step"], "ms/step; roofline", {k: j["roofline"].get(k) for k in ("avg_launch_ms", "avg_ms_alone", "launch_interval_ms", "achieved", "frac")})
for f in glob.glob("gpurun_out/chk/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print("rocprofv3:", r["Name"][:40], r["Calls"], "calls, avg", r["AverageNs"], "ns")
PY
