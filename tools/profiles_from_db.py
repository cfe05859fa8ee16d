#!/usr/bin/env python3
"""Turns the rocpd databases tools/collect_profiles.sh left under gpurun_out/<tag>/ into the small summaries committed
under profiles/: per-kernel stats CSVs (same columns as `rocprofv3 --stats`) and counter JSONs.
usage: profiles_from_db.py gpurun_out/r02 profiles r02"""
import csv
import glob
import json
import os
import sqlite3
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "multimodal_autonomous_driving_perception_and_planning_amd", "csrc")


def stamp(kernel_file):
    """What bench.py needs to decide whether a counter value still describes the kernel it is printed next to: the commit the
    profile was taken at and a hash of the kernel's source file (bench.py drops `traffic` to null when the file has changed)."""
    import hashlib
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    h = hashlib.sha256(open(os.path.join(CSRC, kernel_file), "rb").read()).hexdigest()[:16]
    return {"measured_at_commit": commit, "kernel_source": "csrc/" + kernel_file, "kernel_source_sha256_16": h, "round": tag}


def db_of(name):
    c = glob.glob(os.path.join(src, name, "*_results.db"))
    return sqlite3.connect(c[0]) if c else None


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def stats(name, out):
    db = db_of(name)
    if db is None:
        return
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels "
                      "group by name order by sum(end - start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([short(n), c, int(t), round(a, 1), round(100.0 * t / tot, 2), int(mn), int(mx)])


def counters(name):
    db = db_of(name)
    if db is None:
        return {}
    out = {}
    for k, c, n, v, d in db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                    "group by kernel_name, counter_name"):
        e = out.setdefault(short(k), {"launches": n, "avg_ns": round(d or 0, 1)})
        e[c] = v
    return out


for nm in ("bench_config4", "bench_config4_serial", "bench_config4_window256", "bench_config3", "bench_config2", "lane_S64", "yolo_b64", "yolo_b64_fp32", "yolo_b1_frame"):
    stats(nm, "%s_%s_kernel_stats.csv" % (tag, nm))

# ---- YOLO: launch-by-launch timeline of the last forward of the trace (tools/ytimeline.py) ---------------------------------
_yc = glob.glob(os.path.join(src, "yolo_b64", "*_results.db"))
if _yc:
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    txt = subprocess.run([sys.executable, os.path.join(here, "ytimeline.py"), _yc[0]], capture_output=True, text=True).stdout
    with open(os.path.join(dst, "%s_yolo_b64_timeline.txt" % tag), "w") as f:
        f.write("# one forward of 64 frames, launch by launch (rocprofv3 --kernel-trace -- python3 tools/ybench.py --batch 64 --reps 5)\n" + txt)

# ---- lane: HBM bytes per kernel (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md; both in KiB) -----------------------------
fe, wr, sq = counters("lane_fetch"), counters("lane_write"), counters("lane_sq")
S, H, W = 64, 720, 1280
px = S * H * W
lane = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- python3 tools/lbench.py --reps 2; "
                  "64 frames of 1280x720 per launch; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B)",
        "pixels_per_launch": px, "kernels": {}}
pix_total = 0.0
pixel_kernels = ("front_pack", "front_stream", "thresholds_kernel", "ccl_tile_kernel", "ccl_border_kernel", "finalize_fast", "resolve_bits_kernel",
                 "compact_box_kernel")
for k in sorted(set(fe) | set(wr)):
    if not any(t in k for t in ("front_pack", "front_stream", "thresholds", "ccl_", "finalize", "resolve_bits", "compact", "hough", "lane_fit")):
        continue
    f = 2.0 * fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
    w_ = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    e = {"avg_us": round(fe.get(k, wr.get(k, {})).get("avg_ns", 0) / 1e3, 2), "fetch_bytes": int(f), "write_bytes": int(w_),
         "hbm_bytes_per_px": round((f + w_) / px, 4)}
    if k in sq:
        q = sq[k]
        if q.get("GRBM_GUI_ACTIVE") and q.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
            e["valu_busy_frac"] = round(4.0 * q["SQ_ACTIVE_INST_VALU"] / 1024.0 / (q["GRBM_GUI_ACTIVE"] / 8.0), 4)
            e["SQ_INSTS_VALU"] = q.get("SQ_INSTS_VALU")
            e["wave_cycle_split"] = {m: round(q[m] / q["SQ_WAVE_CYCLES"], 3) for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if q.get("SQ_WAVE_CYCLES")}
    lane["kernels"][k] = e
    if any(k.startswith(t) for t in pixel_kernels):
        pix_total += f + w_
lane["pixel_stage_hbm_bytes_per_px"] = round(pix_total / px, 4)
lane.update(stamp("lane.hip"))
if lane["kernels"]:
    json.dump(lane, open(os.path.join(dst, "lane_pmc.json"), "w"), indent=1)

# ---- planner -----------------------------------------------------------------------------------------------------------------
fe, wr = counters("plan_fetch"), counters("plan_write")
for k in fe:
    if "planner_wave_kernel" in k:
        states = 64 * 256
        f, w_ = 2.0 * fe[k]["FETCH_SIZE"] * 1024.0, wr[k]["WRITE_SIZE"] * 1024.0
        json.dump({"kernel": k, "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, each with --kernel-trace only) -- "
                                          "python3 tools/kbench.py --streams 64 --window 256 --stages plan --reps 2, MI355X, " + tag,
                   "states_per_launch": states, "WRITE_SIZE_KiB": wr[k]["WRITE_SIZE"], "FETCH_SIZE_KiB_raw": fe[k]["FETCH_SIZE"],
                   "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section)", "hbm_bytes_per_launch": int(f + w_),
                   "hbm_bytes_per_state": round((f + w_) / states), "algorithmic_bytes_per_state": 51660, **stamp("planner.hip")},
                  open(os.path.join(dst, "planner_pmc.json"), "w"), indent=2)

# ---- tracker (the headline step's longest kernel: latency-bound, its HBM traffic is reported for completeness) --------------------
fe, wr = counters("trk_fetch"), counters("trk_write")
for k in fe:
    if "tracker_kernel" in k and k in wr:
        frames = 64 * 256
        f, w_ = 2.0 * fe[k]["FETCH_SIZE"] * 1024.0, wr[k]["WRITE_SIZE"] * 1024.0
        json.dump({"kernel": k, "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, each with --kernel-trace only) -- "
                                          "python3 tools/kbench.py --streams 64 --window 256 --stages detect,track --reps 2, MI355X, " + tag,
                   "frames_per_launch": frames, "WRITE_SIZE_KiB": wr[k]["WRITE_SIZE"], "FETCH_SIZE_KiB_raw": fe[k]["FETCH_SIZE"],
                   "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section)", "hbm_bytes_per_launch": int(f + w_),
                   "hbm_bytes_per_frame": round((f + w_) / frames, 1), "algorithmic_bytes_per_frame": 4776, **stamp("tracker.hip")},
                  open(os.path.join(dst, "tracker_pmc.json"), "w"), indent=2)

# ---- the one-launch time-step (headline): HBM bytes per launch ------------------------------------------------------------------
fe, wr = counters("step_fetch"), counters("step_write")
_hs = [k for k in fe if "hot_step_kernel" in k and k in wr]
for k in sorted(_hs, key=lambda k: fe[k]["launches"])[-1:]:           # the instance the headline launches (eight waves per workgroup at depth 4)
    if True:
        f, w_ = 2.0 * fe[k]["FETCH_SIZE"] * 1024.0, wr[k]["WRITE_SIZE"] * 1024.0
        json.dump({"kernel": k, "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, each with --kernel-trace only) -- "
                                          "python3 bench.py --no-also --no-cpu-baseline (64 streams, one launch per time-step, up to 4 steps in flight; "
                                          "the counter passes run the launches one after the other), MI355X, " + tag,
                   "streams_per_launch": 64, "WRITE_SIZE_KiB": wr[k]["WRITE_SIZE"], "FETCH_SIZE_KiB_raw": fe[k]["FETCH_SIZE"],
                   "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section)", "hbm_bytes_per_launch": int(f + w_),
                   "hbm_bytes_per_stream_step": round((f + w_) / 64.0, 1), "algorithmic_bytes_per_stream_step": 4776 + 768 + 228 + 51660,
                   **stamp("step.hip")}, open(os.path.join(dst, "hot_step_pmc.json"), "w"), indent=2)

# ---- YOLO: HBM bytes per kernel family and per forward (FETCH_SIZE x2 + WRITE_SIZE, separate passes) ----------------------------
fe, wr = counters("yolo_fetch"), counters("yolo_write")
if fe and wr:
    fam, tot_f, tot_w = {}, 0.0, 0.0
    # launches per forward: the profiled run does a fixed number of forwards (ybench: 1 warm-up + reps); normalise per forward by
    # the launch count of a kernel that runs exactly once per forward
    nfw = max(1, fe.get("front_fused_kernel", {}).get("launches", 1))
    for k in sorted(set(fe) & set(wr)):
        if not any(t in k for t in ("conv", "c2f", "front_fused", "sppf", "nms", "upsample", "preprocess", "decode", "maxpool")):
            continue
        n = fe[k]["launches"]
        f, w_ = 2.0 * fe[k]["FETCH_SIZE"] * 1024.0 * n / nfw, wr[k]["WRITE_SIZE"] * 1024.0 * n / nfw
        fam[k] = {"launches_per_forward": round(n / nfw, 2), "avg_us": round(fe[k]["avg_ns"] / 1e3, 2), "fetch_bytes_per_forward": int(f),
                  "write_bytes_per_forward": int(w_), "hbm_GBs_while_running": round((f + w_) / max(fe[k]["avg_ns"] * n / nfw, 1.0), 1)}
        tot_f += f
        tot_w += w_
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/ybench.py --batch 64 --reps 2; "
                         "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); bytes per forward of 64 frames",
               "hbm_bytes_per_forward": int(tot_f + tot_w), "fetch_bytes_per_forward": int(tot_f), "write_bytes_per_forward": int(tot_w),
               "kernels": fam, **stamp("yolo.hip")}, open(os.path.join(dst, "yolo_hbm_pmc.json"), "w"), indent=1)

# ---- YOLO: LDS bank conflicts / instruction mix per convolution family -------------------------------------------------------------
ld = counters("yolo_lds")
if ld:
    per = {}
    for k, q in ld.items():
        if not any(t in k for t in ("conv", "c2f", "front_fused")):
            continue
        act, mf_ = q.get("SQ_LDS_IDX_ACTIVE", 0.0), q.get("SQ_INSTS_MFMA", 0.0)
        per[k] = {"launches": q["launches"], "avg_us": round(q["avg_ns"] / 1e3, 2),
                  "lds_conflict_cycles_over_lds_active": round(q.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4) if act else None,
                  "lds_insts_per_mfma": round(q.get("SQ_INSTS_LDS", 0.0) / mf_, 2) if mf_ else None,
                  "valu_insts_per_mfma": round(q.get("SQ_INSTS_VALU", 0.0) / mf_, 2) if mf_ else None}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES -- "
                         "python3 tools/ybench.py --batch 64 --reps 2 (MI355X, " + tag + ")", "kernels": per, **stamp("yolo.hip")},
              open(os.path.join(dst, "yolo_lds_pmc.json"), "w"), indent=1)

# ---- YOLO: MFMA busy ------------------------------------------------------------------------------------------------------------
mf = counters("yolo_mfma")
tot_busy = tot_act = 0.0
per = {}
for k, q in mf.items():
    if not any(t in k for t in ("conv", "c2f", "front_fused")) or not q.get("GRBM_GUI_ACTIVE"):      # every kernel with MFMA convolutions
        continue
    busy, act = q.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * q["launches"], q["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 * q["launches"]
    tot_busy += busy
    tot_act += act
    per[k] = {"launches": q["launches"], "avg_us": round(q["avg_ns"] / 1e3, 2), "mfma_busy_percent": round(100.0 * busy / act, 2)}
if tot_act:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE -- python3 tools/ybench.py --batch 64 --reps 3; "
                         "busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
               "overall_mfma_busy_percent": round(100.0 * tot_busy / tot_act, 2), "kernels": per, **stamp("yolo.hip")},
              open(os.path.join(dst, "yolo_mfma_pmc.json"), "w"), indent=1)
print("profiles written to", dst)
