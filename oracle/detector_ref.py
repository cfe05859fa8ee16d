"""Oracle D1: simulated detector (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates ObjectDetector._detect_simulated (src/perception/detector.py:125-169)
as a pure function of (frame_count, h, w).  The reference reseeds NumPy's
global legacy MT19937 stream with `frame_count % 1000` on every call
(detector.py:134), so a private RandomState seeded the same way yields the
same draws.  Draw order per vehicle (detector.py:141-166): uniform, randint,
randint, choice(p), uniform.

Pinned bit-for-bit by tests/golden/detections.npz.
"""
import numpy as np

CLASS_NAMES = ("car", "truck", "pedestrian", "cyclist", "motorcycle", "bus",
               "traffic_light", "stop_sign")                       # detector.py:39-48
CLASS_WEIGHTS = (0.6, 0.15, 0.1, 0.05, 0.03, 0.05, 0.01, 0.01)     # detector.py:159
DMAX = 8


def simulated_detections(frame_count, h, w):
    """-> (n, box int32[n,4], cls int32[n], conf float64[n]) for one frame."""
    rs = np.random.RandomState(frame_count % 1000)
    n = int(rs.randint(3, 8))
    box = np.zeros((n, 4), np.int32)
    cls = np.zeros(n, np.int32)
    conf = np.zeros(n, np.float64)
    t = frame_count * 0.02
    for i in range(n):
        depth = rs.uniform(0.3, 1.0)
        bw = int(80 * depth + 40)
        bh = int(60 * depth + 30)
        xb = (i * 150 + int(50 * np.sin(t + i))) % (w - bw)
        yb = int(h * 0.4 + (h * 0.4 * depth))
        x1 = max(0, xb + int(rs.randint(-10, 10)))
        y1 = max(0, yb + int(rs.randint(-5, 5)))
        box[i] = (x1, y1, min(w, x1 + bw), min(h, y1 + bh))
        cls[i] = rs.choice(len(CLASS_WEIGHTS), p=CLASS_WEIGHTS)
        conf[i] = rs.uniform(0.75, 0.98)
    return n, box, cls, conf


def detection_table(first_frame_count, n_frames, h, w, dcap=DMAX):
    """Table for frame_count = first .. first+n_frames-1 (the value AFTER detect()'s increment)."""
    n = np.zeros(n_frames, np.int32)
    box = np.zeros((n_frames, dcap, 4), np.int32)
    cls = np.zeros((n_frames, dcap), np.int32)
    conf = np.zeros((n_frames, dcap), np.float64)
    for f in range(n_frames):
        k, b, c, p = simulated_detections(first_frame_count + f, h, w)
        n[f] = k
        box[f, :k], cls[f, :k], conf[f, :k] = b, c, p
    return n, box, cls, conf


def centers(box):
    """Detection.__post_init__ (detector.py:23-26): ((x1+x2)/2, (y1+y2)/2) in float64."""
    b = np.asarray(box, np.float64)
    return np.stack([(b[..., 0] + b[..., 2]) / 2, (b[..., 1] + b[..., 3]) / 2], axis=-1)
