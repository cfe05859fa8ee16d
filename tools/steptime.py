#!/usr/bin/env python3
"""Where an overlapped time-step goes: 100-MHz clock stamps of each role's thread 0, averaged over all launches and streams
(AVHOT_STEP_FENCE=8 makes the step kernel collect them in the tail of seq_flags).  usage: python tools/steptime.py [depth]"""
import os, sys, time
os.environ["AVHOT_STEP_FENCE"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
S, D, N = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 4000
lp = HotLoop(n_streams=S, window=1, overlap=D)
print("stream sets tried:", lp.tune_streams())
lp.load_measurements(np.stack([np.asarray(generate_ego_motion(64, seed=s), np.float64)[:1] for s in range(S)]), all_sets=True)
lp.enqueue_steps(200); lp.synchronize()
base = 65 * S + 32
lp.seq_flags[base:base + 64].zero_(); torch.cuda.synchronize()
t0 = time.perf_counter(); lp.enqueue_steps(N); lp.synchronize(); dt = (time.perf_counter() - t0) / N * 1e6
st = lp.seq_flags[base:base + 64].cpu().numpy().view(np.uint64).reshape(2, 16).astype(np.float64)
print("depth %d: %.2f us per step (with the clock stamps)" % (D, dt))
names = [["detections", "wait for predecessor", "record -> LDS (+barrier)", "tracker frame, record out + acknowledged, counter, outputs", "-", "-", "-"],
         ["-", "wait for predecessor", "record -> LDS", "Kalman + record out", "barrier", "-", "publish + planner"]]
for r, role in enumerate(("tracker role", "Kalman / planner role")):
    n = st[r, 15]
    print(role, "(%d launches x streams)" % n)
    for k in range(7):
        if names[r][k] != "-":
            print("   %-32s %7.2f us" % (names[r][k], st[r, k] / n * 0.01))
    if st[r, 9]:
        print("   %-32s %7.2f us   (publisher's counter store -> consumer's poll returned; included in the wait)" % ("hand-over latency", st[r, 8] / st[r, 9] * 0.01))
