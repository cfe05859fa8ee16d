#!/usr/bin/env python3
"""Reads bench.py's JSON line on stdin and prints value, ms/step and the per-stage times (tuning aid)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, d["value"], d["ms_per_step"], [(k["stage"], k["avg_ms"]) for k in d["kernels"]])
