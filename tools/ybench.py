#!/usr/bin/env python3
"""YOLO-mode detector timing (batch of frames per launch)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
from multimodal_autonomous_driving_perception_and_planning_amd.perception.yolo import YoloV8n
from multimodal_autonomous_driving_perception_and_planning_amd.harness import synthetic_frame
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
a = ap.parse_args()
B = a.batch
m = YoloV8n("random:0", batch=B, precision=a.precision); m._prepare(720, 1280)
fr = torch.as_tensor(np.stack([synthetic_frame(720, 1280, s % 4, 0) for s in range(B)])).cuda()
L = nat.lib(); st = m._dev.stream
ea, eb = C.c_void_p(), C.c_void_p(); L.av_event_create(C.byref(ea)); L.av_event_create(C.byref(eb))
m.forward_device(fr); torch.cuda.synchronize()
best = 1e9; ms = C.c_float()
for _ in range(a.reps):
    L.av_event_record(ea, st); m.forward_device(fr); L.av_event_record(eb, st); L.av_event_elapsed_ms(ea, eb, C.byref(ms)); best = min(best, ms.value)
gf = 5.2 * B
print(a.precision, "yolo batch=%d: %.3f ms  %.1f frames/s  ~%.1f TFLOP/s (5.2 GFLOP/frame)" % (B, best, B / best * 1e3, gf / best))
