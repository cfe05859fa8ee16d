#!/usr/bin/env python3
"""The headline's step kernel (HotLoop(64, window 1, overlap=4): eight waves per workgroup, sequence counters, device-scope hand-over)
launched ONE AT A TIME, the host waiting after every launch: the form a counter pass needs.  rocprofv3 --pmc runs kernels one after
the other in an order of its own choosing, and a step whose predecessor has not run yet waits for it -- with the launches overlapped as
bench.py runs them the pass ends in the loop's fault word instead of counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
S = 64
lp = HotLoop(n_streams=S, window=1, overlap=4)
lp.reset(frame_offsets=[17 * s for s in range(S)])
lp.load_measurements(np.stack([np.asarray(generate_ego_motion(64, seed=s), np.float64)[:1] for s in range(S)]), all_sets=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    lp.enqueue_step()
    lp.synchronize()
print("done", lp.frame_count.cpu().numpy()[:4])
