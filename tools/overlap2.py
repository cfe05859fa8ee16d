#!/usr/bin/env python3
"""Upper bound for overlapping consecutive time-steps: K independent 64-stream loops (own HIP stream each) stepped in turn by one
host thread.  Device time per launch when K launches may be in flight against the one-at-a-time 13.4 us."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
S = 64
z = np.stack([np.asarray(generate_ego_motion(64, seed=s), np.float64)[:1] for s in range(S)])
for K in (() if os.environ.get("SKIPK") else (1, 2, 3, 4)):
    loops = [HotLoop(n_streams=S, window=1) for _ in range(K)]
    for lp in loops:
        lp.load_measurements(z)
    for _ in range(200):
        for lp in loops: lp.enqueue_step()
    for lp in loops: lp.synchronize()
    N = 4000
    t0 = time.perf_counter()
    for i in range(N): loops[i % K].enqueue_step()
    t1 = time.perf_counter()
    for lp in loops: lp.synchronize()
    t2 = time.perf_counter()
    print("K=%d loops in turn: host enqueue %.2f us/launch, until the device is done %.2f us/launch" % (K, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
    del loops

for ov in (1, 2, 3):
    lp = HotLoop(n_streams=S, window=1, overlap=ov)
    lp.load_measurements(z, **({"all_sets": True} if ov == 2 else {}))
    for _ in range(200): lp.enqueue_step()
    lp.synchronize()
    N = 4000
    t0 = time.perf_counter()
    for _ in range(N): lp.enqueue_step()
    t1 = time.perf_counter()
    lp.synchronize()
    t2 = time.perf_counter()
    print("HotLoop(overlap=%d): host enqueue %.2f us/step, until the device is done %.2f us/step" % (ov, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)

for D in (2, 3, 4):
    lp = HotLoop(n_streams=S, window=1, overlap=D)
    if os.environ.get("TUNE"): print("  stream sets tried, us/step:", lp.tune_streams())
    lp.load_measurements(z, all_sets=True)
    lp.enqueue_steps(200); lp.synchronize()
    for N in (4000,):
        t0 = time.perf_counter()
        lp.enqueue_steps(N)
        t1 = time.perf_counter()
        lp.synchronize()
        t2 = time.perf_counter()
        print("HotLoop(overlap=%d).enqueue_steps(%d): host enqueue %.2f us/step, until the device is done %.2f us/step" % (D, N, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
    del lp
