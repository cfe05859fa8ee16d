#!/usr/bin/env python3
"""Per-stage kernel timings (HIP events on the launch stream) for tuning.  Not part of the bench contract."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat  # noqa: E402
from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop  # noqa: E402
from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion  # noqa: E402


def timeit(loop, fn, reps):
    L = nat.lib()
    a, b = C.c_void_p(), C.c_void_p()
    nat.check(L.av_event_create(C.byref(a)))
    nat.check(L.av_event_create(C.byref(b)))
    fn()
    loop.synchronize()
    best, tot = 1e9, 0.0
    ms = C.c_float()
    for _ in range(reps):
        nat.check(L.av_event_record(a, loop._s))
        fn()
        nat.check(L.av_event_record(b, loop._s))
        nat.check(L.av_event_elapsed_ms(a, b, C.byref(ms)))
        best = min(best, ms.value)
        tot += ms.value
    return best, tot / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--window", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tcap", type=int, default=64)
    ap.add_argument("--stages", default="detect,track,kf,plan")   # also: maneuver, interact
    ap.add_argument("--no-wp", action="store_true", help="planner: costs/order only (no waypoint stores)")
    a = ap.parse_args()
    S, W = a.streams, a.window
    loop = HotLoop(n_streams=S, window=W, tcap=a.tcap, keep_waypoints=not a.no_wp)
    loop.reset(frame_offsets=[s * 17 for s in range(S)])
    loop.load_measurements(np.stack([np.asarray(generate_ego_motion(W, seed=s % 8), np.float64) for s in range(S)]))
    loop.step(sync=True)
    F = S * W
    for name in a.stages.split(","):
        fn = {"detect": loop.enqueue_detect, "track": loop.enqueue_track, "kf": loop.enqueue_kf,
              "plan": loop.enqueue_plan, "maneuver": loop.enqueue_maneuver, "interact": loop.enqueue_interactions}[name]
        best, avg = timeit(loop, fn, a.reps)
        line = "%-7s S=%d W=%d  best %.4f ms  avg %.4f ms  %.3f us/frame-step  %.2f Mframes/s" % (
            name, S, W, best, avg, best * 1e3 / W, F / best / 1e3)
        if name == "plan":
            gb = loop.planner_bytes_per_state() * F / 1e9
            line += "  %.0f GB/s (%.1f%% of 8 TB/s)" % (gb / (best * 1e-3), gb / (best * 1e-3) / 80.0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
