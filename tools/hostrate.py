#!/usr/bin/env python3
"""Host time per enqueued time-step (HotLoop window 1) against the device time per step: is the one-launch loop host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
for S in (64, 1):
    loop = HotLoop(n_streams=S, window=1)
    loop.load_measurements(np.stack([np.asarray(generate_ego_motion(64, seed=s), np.float64)[:1] for s in range(S)]))
    for _ in range(200): loop.enqueue_step()
    loop.synchronize()
    N = 4000
    t0 = time.perf_counter()
    for _ in range(N): loop.enqueue_step()
    t1 = time.perf_counter()
    loop.synchronize()
    t2 = time.perf_counter()
    print("S=%d: host enqueue %.2f us/step, until the device is done %.2f us/step" % (S, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
