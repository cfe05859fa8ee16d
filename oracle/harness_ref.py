"""Oracle H: the per-frame demo loop on the CPU (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates the call order of demo.py:97-120 (detect -> track -> KF step -> plan;
plan() is called with the tuple (state.x, state.y, state.heading, state.speed)
and no obstacles) and VideoDataLoader.generate_ego_motion
(data/loaders/video_loader.py:166-205).  The reference draws the measurement
noise from the unseeded global NumPy stream; the harness seeds a private legacy
RandomState so GPU and CPU runs see identical inputs.

Used by tests and by bench.py's cpu_baseline leg ("port").
"""
import time

import numpy as np

from .detector_ref import detection_table
from .kf_ref import KalmanRef
from .planner_ref import PlannerRef
from .tracker_ref import TrackerRef


def ego_motion(num_steps, fps=30.0, seed=0):
    """-> float64[num_steps,4] rows (x, y, vx, vy)  (video_loader.py:184-203)."""
    rs = np.random.RandomState(seed)
    dt = 1.0 / fps
    out = np.zeros((num_steps, 4))
    x = y = 0.0
    speed = 10.0
    for i in range(num_steps):
        t = i * dt
        heading = 0.05 * np.sin(t * 0.5)
        vx = speed * np.cos(heading)
        vy = speed * np.sin(heading)
        x += vx * dt
        y += vy * dt
        out[i] = (x + rs.normal(0, 0.1), y + rs.normal(0, 0.1),
                  vx + rs.normal(0, 0.05), vy + rs.normal(0, 0.05))
    return out


def run_stream(n_frames, h=720, w=1280, frame_offset=0, ego_seed=0, tcap=64, keep=True):
    """One stream of the simulated-detection loop.  Returns per-frame arrays."""
    n, box, cls, conf = detection_table(frame_offset + 1, n_frames, h, w)
    z = ego_motion(n_frames, seed=ego_seed)
    trk, kf, pl = TrackerRef(), KalmanRef(), PlannerRef()
    out = dict(det_n=n, det_box=box, det_cls=cls, det_conf=conf, z=z)
    if keep:
        out.update(ids=np.full((n_frames, tcap), -1, np.int32), n_live=np.zeros(n_frames, np.int32),
                   tbox=np.zeros((n_frames, tcap, 4), np.int32), ahm=np.zeros((n_frames, tcap, 3), np.int32),
                   det2trk=np.full((n_frames, box.shape[1]), -1, np.int32),
                   state=np.zeros((n_frames, 12)), cost=np.zeros((n_frames, 21)),
                   order=np.zeros((n_frames, 21), np.int32), best_wp=np.zeros((n_frames, 51, 6)))
    for f in range(n_frames):
        r = trk.update(n[f], box[f], cls[f], conf[f])
        st = kf.step(z[f])
        p = pl.plan((st[0], st[1], st[4], st[5]))
        if keep:
            t = trk.table(tcap)
            out["n_live"][f] = t["n"]
            out["ids"][f] = t["ids"]
            out["tbox"][f] = t["box"]
            out["ahm"][f] = t["ahm"]
            out["det2trk"][f, :n[f]] = r["det2trk"]
            out["state"][f] = st
            out["cost"][f] = p["cost"]
            out["order"][f] = p["order"]
            out["best_wp"][f] = p["wp"][p["order"][0]]
    return out


def time_cpu_loop(n_frames, warmup=5, h=720, w=1280):
    """Frames/s of the single-thread CPU loop, detector included (it is part of the per-frame cost)."""
    from .detector_ref import simulated_detections

    z = ego_motion(n_frames + warmup)
    trk, kf, pl = TrackerRef(), KalmanRef(), PlannerRef()
    t0 = None
    for f in range(n_frames + warmup):
        if f == warmup:
            t0 = time.perf_counter()
        k, b, c, p = simulated_detections(f + 1, h, w)
        trk.update(k, b, c, p)
        st = kf.step(z[f])
        pl.plan((st[0], st[1], st[4], st[5]))
    return n_frames / (time.perf_counter() - t0)
