#!/usr/bin/env python3
"""Per-frame YOLO-mode detector call (ObjectDetector(mode="yolo").detect(frame), demo.py:107): median ms with the forward replayed
as one hipGraph and with eager launches (AVHOT_YOLO_NO_GRAPH=1), and where the time goes (upload / device / wait)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multimodal_autonomous_driving_perception_and_planning_amd.harness import synthetic_frame
from multimodal_autonomous_driving_perception_and_planning_amd.perception import yolo as Y

imgs = [synthetic_frame(720, 1280, 0, f) for f in range(8)]
for mode in ("graph", "eager"):
    if mode == "eager":
        os.environ["AVHOT_YOLO_NO_GRAPH"] = "1"
    m = Y.YoloV8n("random:0")
    ts = []
    for i in range(120):
        t0 = time.perf_counter()
        m.detect(imgs[i % 8])
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts[20:])
    # the upload alone
    tu = []
    for i in range(50):
        t0 = time.perf_counter()
        m._io_in.upload_from("frame", imgs[i % 8])
        m._gdev.sync()
        tu.append((time.perf_counter() - t0) * 1e3)
    print("%s: detect median %.4f ms  p10 %.4f  p90 %.4f | frame upload alone %.4f ms" % (mode, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), np.median(tu)), flush=True)
    m.close()
