#!/usr/bin/env python3
"""Fused vs separate launches, element by element: where do the outputs of a tapped layer differ?
usage: yfuse_dbg.py [tensor id, default 4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd.perception.yolo import YoloV8n
from multimodal_autonomous_driving_perception_and_planning_amd.harness import synthetic_frame
tid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 4
rs = np.random.RandomState(11)
frames = [synthetic_frame(720, 1280, s, 3 * s) for s in range(B - 1)] + [rs.randint(0, 256, (720, 1280, 3)).astype(np.uint8)]
m = YoloV8n("random:0", batch=B); m._prepare(720, 1280)
m._frames.copy_(torch.as_tensor(np.stack(frames)))
def run():
    m.forward_device(m._frames); torch.cuda.synchronize()
    return m.tensor(tid, image=None)
f = run()
os.environ["AVHOT_YOLO_NO_FUSE"] = "1"
u = run()
d = f != u
print("tensor %d shape %s: %d of %d elements differ, max |d| %.3g" % (tid, f.shape, d.sum(), d.size, np.abs(f - u).max()))
if d.any():
    n, y, x, c = np.nonzero(d)
    print("per image:", np.bincount(n, minlength=B))
    print("rows  (y %% 16):", np.bincount(y % 16, minlength=16))
    print("cols  (x %% 16):", np.bincount(x % 16, minlength=16))
    print("chans (c %% 16):", np.bincount(c % 16, minlength=16), " c // 16:", np.bincount(c // 16))
    print("y hist:", np.bincount(y, minlength=f.shape[1]))
    print("x hist:", np.bincount(x, minlength=f.shape[2]))
    print("c // 32:", np.bincount(c // 32))
    i = np.argmax(np.abs(f - u)); print("worst:", np.unravel_index(i, f.shape), f.flat[i], u.flat[i])
