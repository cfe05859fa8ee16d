#!/usr/bin/env python3
"""Occupancy audit: registers, scratch, static LDS and the compiler's waves-per-SIMD figure of every kernel of csrc/*.hip
(hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).

Why it exists: `__launch_bounds__(threads, n)` promises n waves per SIMD, not n workgroups per CU.  c2f16_fused_kernel was
written for two 512-thread workgroups per CU and declared (512, 2); the compiler took 132 registers and the kernel ran one
workgroup per CU for a round (109 us instead of 83).  Read the table against the residency each kernel's design assumes:
waves per CU = 4 x waves/SIMD, workgroups per CU = min(that / waves per workgroup, 160 KB / LDS per workgroup).

usage: python tools/occupancy.py [file.hip ...]      (default: every csrc/*.hip)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal_autonomous_driving_perception_and_planning_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
FIELDS = (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize \\[bytes/lane\\]", "scratch"), ("Occupancy \\[waves/SIMD\\]", "occ"),
          ("LDS Size \\[bytes/block\\]", "lds"), ("VGPRs Spill", "spill"))
print("%-12s %-78s %5s %5s %7s %5s %8s" % ("file", "kernel", "vgpr", "spill", "scratch", "w/SIMD", "staticLDS"))
for f in files:
    if os.path.basename(f) == "step.hip":           # includes the stage files: same device code, one extra kernel
        pass
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                        "-I", CSRC, "-c", f, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur, rows = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows.setdefault(cur, {})
            continue
        for pat, key in FIELDS:
            m = re.search(r"remark:\s+" + pat + r": (\d+)", line)
            if m and cur is not None and key not in rows[cur]:
                rows[cur][key] = int(m.group(1))
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%-12s %-78s %5s %5s %7s %5s %8s" % (os.path.basename(f), name[:78], v.get("vgpr", ""), v.get("spill", ""), v.get("scratch", ""),
                                                   v.get("occ", ""), v.get("lds", "")))
