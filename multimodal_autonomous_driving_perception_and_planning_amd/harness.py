"""Demo-style harness: the reference's per-frame loop (demo.py:97-120) over the drop-in classes.

`python -m ...harness --test` re-creates the self test the reference's README advertises
(README.md:169-187, six checks; the flag itself does not exist in the reference's demo.py)
on synthetic 1280x720 input, then runs the 300-frame loop and prints frames/s.
"""
import argparse
import sys
import time

import numpy as np


def generate_ego_motion(num_steps, fps=30.0, seed=0):
    """Synthetic [x, y, vx, vy] measurements: VideoDataLoader.generate_ego_motion's formula
    (data/loaders/video_loader.py:184-203) with an explicit seed instead of the global RNG."""
    rs = np.random.RandomState(seed)
    dt = 1.0 / fps
    out = []
    x = y = 0.0
    for i in range(num_steps):
        hd = 0.05 * np.sin(i * dt * 0.5)
        vx, vy = 10.0 * np.cos(hd), 10.0 * np.sin(hd)
        x += vx * dt
        y += vy * dt
        out.append((x + rs.normal(0, 0.1), y + rs.normal(0, 0.1), vx + rs.normal(0, 0.05), vy + rs.normal(0, 0.05)))
    return out


def synthetic_frame(h=720, w=1280, stream=0, frame=0):
    """Host-side synthetic road frame (same integer formulas as the oracle's generator)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    hz = (h * 9) // 20
    img = np.zeros((h, w, 3), np.int64)
    sky = y < hz
    for c, (a, b) in enumerate(((230, 60), (190, 50), (150, 70))):
        img[..., c] = np.where(sky, a - (y * b) // max(hz, 1), 0)
    hsh = ((x * 73856093) ^ (y * 19349663) ^ ((stream * 83492791 + frame * 2654435761) & 0xFFFFFFFF)) & 0xFFFFFFFF
    hsh = ((hsh ^ (hsh >> 13)) * 1274126177) & 0xFFFFFFFF
    tex = (hsh >> 24) & 15
    for c in range(3):
        img[..., c] = np.where(~sky, 84 + tex + (2 - c) * 2, img[..., c])
    den, t = max(h - hz, 1), y - hz
    sway = ((stream * 7 + frame) % 32) - 16
    for (xt, xb) in (((w * 9) // 20, (w * 3) // 20), ((w * 11) // 20, (w * 17) // 20)):
        xc = xt + ((xb - xt + sway) * t) // den
        on = (~sky) & ((((y + 5 * frame) // 24) % 2) == 0) & (np.abs(x - xc) <= 1 + (6 * t) // den)
        for c in range(3):
            img[..., c] = np.where(on, 235, img[..., c])
    from .generators import vehicle_boxes
    for x1, y1, x2, y2, col in vehicle_boxes(h, w, stream, frame):
        img[y1:y2, x1:x2] = col
    return img.astype(np.uint8)


def run_loop(num_frames=300, h=720, w=1280, lanes=True, verbose=True):
    from .perception import LaneDetector, ObjectDetector
    from .planning import MotionPlanner
    from .state_estimation import VehicleStateEstimator
    from .tracking import MultiObjectTracker

    detector, lane_detector = ObjectDetector(mode="simulated"), LaneDetector()
    tracker, estimator, planner = MultiObjectTracker(), VehicleStateEstimator(), MotionPlanner()
    ego = generate_ego_motion(num_frames)
    frames = [synthetic_frame(h, w, 0, f) for f in range(min(num_frames, 8))]
    t0 = time.perf_counter()
    for i in range(num_frames):
        frame = frames[i % len(frames)]
        detections = detector.detect(frame)
        left, right = lane_detector.detect(frame) if lanes else (None, None)
        tracks = tracker.update(detections)
        state = estimator.step(np.array(ego[i]))
        optimal, candidates = planner.plan((state.x, state.y, state.heading, state.speed))
        if verbose and (i + 1) % 50 == 0:
            fps = (i + 1) / (time.perf_counter() - t0)
            print("Frame %d/%d | FPS: %.1f | Tracks: %d | Speed: %.1f km/h | lanes: %s/%s | best cost %.2f" % (
                i + 1, num_frames, fps, len(tracks), state.speed * 3.6, left is not None, right is not None, optimal.cost))
    return num_frames / (time.perf_counter() - t0)


def self_test():
    from .perception import LaneDetector, ObjectDetector
    from .planning import MotionPlanner
    from .state_estimation import VehicleStateEstimator
    from .tracking import MultiObjectTracker
    frame = synthetic_frame()
    det = ObjectDetector(mode="simulated")
    dets = det.detect(frame)
    assert 3 <= len(dets) <= 7
    print("[Test 1] Object Detector ... %d detections  ✓" % len(dets))
    left, right = LaneDetector().detect(frame)
    assert left is not None and right is not None
    print("[Test 2] Lane Detector ... left/right lanes found  ✓")
    trk = MultiObjectTracker()
    for _ in range(3):
        tracks = trk.update(det.detect(frame))
    print("[Test 3] Multi-Object Tracker ... %d live, %d confirmed  ✓" % (len(trk.tracks), len(tracks)))
    est = VehicleStateEstimator()
    st = None
    for z in generate_ego_motion(10):
        st = est.step(np.array(z))
    assert abs(st.speed - 10.0) < 3.0
    print("[Test 4] Vehicle State Estimator ... speed %.2f m/s  ✓" % st.speed)
    opt, cands = MotionPlanner().plan((st.x, st.y, st.heading, st.speed))
    assert len(cands) == 21 and len(opt.waypoints) == 51
    print("[Test 5] Motion Planner ... %d candidates, best cost %.2f  ✓" % (len(cands), opt.cost))
    from .visualization import BEVRenderer
    panel = BEVRenderer().render(ego_state=st, tracks=tracks, planned_trajectory=opt, candidate_trajectories=cands[:10])
    assert panel.shape == (600, 600, 3) and (panel == (0, 255, 0)).all(axis=2).any()
    print("[Test 6] BEV Renderer ... %dx%d panel rendered on the device  \u2713" % (panel.shape[1], panel.shape[0]))
    fps = run_loop(300, verbose=False)
    print("300-frame loop (1280x720, simulated detection + lane + track + KF + plan, per-frame class API): %.1f FPS" % fps)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--test", action="store_true")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--no-lanes", action="store_true")
    a = ap.parse_args(argv)
    if a.test:
        self_test()
    else:
        print("Average FPS: %.1f" % run_loop(a.frames, lanes=not a.no_lanes))


if __name__ == "__main__":
    sys.exit(main())
