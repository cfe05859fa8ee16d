"""Oracle L1-L7: lane detector (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED for the OpenCV calls (cv2 is not installed; see oracle/c/lane_ref.c for what is
restated and from where).  The NumPy part (L5-L7: slope split, np.polyfit, EMA, resampling) follows
src/perception/lane_detector.py:105-176,178-218 with the real NumPy functions.
"""
import ctypes as C
import os
import subprocess
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblane_ref.so")
_lib = None
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "c")], check=True)
        L = C.CDLL(_SO)
        L.lane_gray.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.lane_blur5.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.lane_median.argtypes = [u8p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lane_median.restype = C.c_double
        L.lane_canny.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.lane_apply_roi.argtypes = [u8p, C.c_int, C.c_int]
        L.lane_roi_bounds.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lane_houghp.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        L.lane_houghp.restype = C.c_int
        _lib = L
    return _lib


def gray(bgr):
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().lane_gray(np.ascontiguousarray(bgr), h, w, out)
    return out


def blur5(g):
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().lane_blur5(np.ascontiguousarray(g), h, w, out)
    return out


def median_thresholds(img):
    lo, hi = C.c_int(), C.c_int()
    med = lib().lane_median(np.ascontiguousarray(img), img.size, C.byref(lo), C.byref(hi))
    assert med == float(np.median(img))
    return med, lo.value, hi.value


def canny(img, lo, hi):
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().lane_canny(np.ascontiguousarray(img), h, w, lo, hi, out)
    return out


def apply_roi(edges):
    out = np.ascontiguousarray(edges).copy()
    lib().lane_apply_roi(out, out.shape[0], out.shape[1])
    return out


def houghp(edges, threshold=50, min_line_length=50, max_line_gap=150, max_lines=4096):
    h, w = edges.shape
    lines = np.zeros((max_lines, 4), np.int32)
    n = lib().lane_houghp(np.ascontiguousarray(edges), h, w, threshold, min_line_length, max_line_gap, max_lines, lines)
    return lines[:n].copy()


def separate(lines, width):
    """lane_detector.py:105-134 -> (left, right) lists of (x1,y1,x2,y2)."""
    left, right = [], []
    cx = width / 2
    for x1, y1, x2, y2 in lines:
        x1, y1, x2, y2 = int(x1), int(y1), int(x2), int(y2)
        if x2 == x1:
            continue
        slope = (y2 - y1) / (x2 - x1)
        if abs(slope) < 0.3:
            continue
        mid = (x1 + x2) / 2
        if slope < 0 and mid < cx:
            left.append((x1, y1, x2, y2))
        elif slope > 0 and mid > cx:
            right.append((x1, y1, x2, y2))
    return left, right


def fit(lines, height, prev, smoothing=0.7):
    """lane_detector.py:136-176 -> None or (points int32[50,2], confidence, coeffs float64[3])."""
    if not lines:
        return None
    xs, ys = [], []
    for x1, y1, x2, y2 in lines:
        xs += [x1, x2]
        ys += [y1, y2]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            co = np.polyfit(ys, xs, 2)
        except np.linalg.LinAlgError:
            return None
    if prev is not None:
        co = smoothing * prev + (1 - smoothing) * co
    yp = np.linspace(height * 0.6, height, 50)
    xp = np.polyval(co, yp)
    pts = np.column_stack((xp, yp)).astype(np.int32)
    return pts, min(1.0, len(lines) / 10), co


class LaneRef:
    def __init__(self):
        self.prev_left = None
        self.prev_right = None

    def stages(self, bgr):
        g = gray(bgr)
        b = blur5(g)
        med, lo, hi = median_thresholds(b)
        e = canny(b, lo, hi)
        m = apply_roi(e)
        segs = houghp(m)
        return dict(gray=g, blur=b, median=med, lo=lo, hi=hi, edges=e, masked=m, segments=segs)

    def detect(self, bgr):
        st = self.stages(bgr)
        h, w = bgr.shape[:2]
        left, right = separate(st["segments"], w)
        L = fit(left, h, self.prev_left)
        R = fit(right, h, self.prev_right)
        if L is not None:
            self.prev_left = L[2]
        if R is not None:
            self.prev_right = R[2]
        st.update(left=L, right=R, n_left=len(left), n_right=len(right))
        return st


def synthetic_frame(h, w, stream=0, frame=0):
    """Deterministic road scene in pure integer arithmetic (the device generator reproduces it bit for bit).

    Sky gradient above the horizon, textured asphalt below, two dashed lane lines converging towards a
    vanishing point, a few box "vehicles".  Stands in for the reference's lost SyntheticDataGenerator
    (SURVEY.md F2).
    """
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    hz = (h * 9) // 20
    img = np.zeros((h, w, 3), np.int64)
    sky = y < hz
    img[..., 0] = np.where(sky, 230 - (y * 60) // max(hz, 1), 0)
    img[..., 1] = np.where(sky, 190 - (y * 50) // max(hz, 1), 0)
    img[..., 2] = np.where(sky, 150 - (y * 70) // max(hz, 1), 0)
    hsh = ((x * 73856093) ^ (y * 19349663) ^ ((stream * 83492791 + frame * 2654435761) & 0xFFFFFFFF)) & 0xFFFFFFFF
    hsh = ((hsh ^ (hsh >> 13)) * 1274126177) & 0xFFFFFFFF
    tex = (hsh >> 24) & 15
    road = ~sky
    for c in range(3):
        img[..., c] = np.where(road, 84 + tex + (2 - c) * 2, img[..., c])
    den = max(h - hz, 1)
    t = y - hz
    sway = ((stream * 7 + frame) % 32) - 16
    for (xt, xb) in (((w * 9) // 20, (w * 3) // 20), ((w * 11) // 20, (w * 17) // 20)):
        xc = xt + ((xb - xt + sway) * t) // den
        half = 1 + (6 * t) // den
        dash = (((y + 5 * frame) // 24) % 2) == 0
        on = road & dash & (np.abs(x - xc) <= half)
        for c in range(3):
            img[..., c] = np.where(on, 235, img[..., c])
    k = 3 + ((stream * 5 + frame // 8) % 4)
    for i in range(k):
        s = (stream * 131 + i * 977 + (frame // 8) * 31) & 0xFFFF
        bw, bh = 50 + (s % 90), 36 + ((s >> 3) % 60)
        bx = (s * 37 + i * 211 + frame * (3 + i)) % max(w - bw, 1)
        by = hz + 10 + ((s >> 5) % max(h - hz - bh - 10, 1))
        on = (x >= bx) & (x < bx + bw) & (y >= by) & (y < by + bh)
        col = (40 + (s % 160), 40 + ((s >> 4) % 160), 40 + ((s >> 8) % 160))
        for c in range(3):
            img[..., c] = np.where(on, col[c], img[..., c])
    return img.astype(np.uint8)
