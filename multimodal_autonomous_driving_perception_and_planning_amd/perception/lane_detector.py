"""LaneDetector -- drop-in surface of src/perception/lane_detector.py over libavhot.so.

detect() uploads the frame and runs the whole chain on the GPU (av_lane_detect): gray, blur, median,
Canny (NMS + union-find hysteresis), ROI, progressive probabilistic Hough, slope split, quadratic fit
with EMA smoothing, 50-point resampling.  The previous smoothed fits (prev_left_fit/prev_right_fit)
live in an 8-double device record that the kernel updates.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev, Packed


@dataclass
class LaneLine:
    """A fitted lane line (lane_detector.py:13-19)."""
    points: np.ndarray
    side: str
    confidence: float
    polynomial: Optional[np.ndarray] = None


def _roi_rows(vertices, h, w):
    """Inclusive column range per row for a convex ROI polygon given as in the reference ([[(x,y),...]])."""
    v = np.asarray(vertices, np.float64).reshape(-1, 2)
    rows = np.zeros((h, 2), np.int32)
    rows[:, 0], rows[:, 1] = 1, 0
    n = len(v)
    for y in range(h):
        xs = []
        for k in range(n):
            (xa, ya), (xb, yb) = v[k], v[(k + 1) % n]
            if ya == yb:
                if ya == y:
                    xs += [xa, xb]
                continue
            t = (y - ya) / (yb - ya)
            if 0.0 <= t <= 1.0:
                xs.append(xa + t * (xb - xa))
        if xs:
            rows[y] = (int(np.floor(min(xs) + 0.5)), int(np.floor(max(xs) + 0.5)))
    return rows


class LaneDetector:
    MAX_SEGMENTS = 512

    def __init__(self, roi_vertices: Optional[np.ndarray] = None, device: int = 0):
        self.roi_vertices = roi_vertices
        self.smoothing_factor = 0.7
        self._dev = Dev(device)
        d = self._dev
        # EMA record and every result live in one host-mapped buffer: the fit kernel reads / writes it in place
        self._io = Packed(d, [("state", np.float64, (1, 8)), ("poly", np.float64, (1, 2, 3)), ("pts", np.int32, (1, 2, 50, 2)),
                              ("info", np.int32, (1, 8)), ("conf", np.float64, (1, 2))])
        self._shape = None
        self._ws = None
        self._roi = None
        self._stage = None

    # prev_*_fit mirror the device record so user code that reads or clears them keeps working
    def _get_fit(self, side):
        rec = self._io.h["state"][0, side * 4:side * 4 + 4]
        return rec[:3].copy() if rec[3] != 0.0 else None

    def _set_fit(self, side, value):
        rec = self._io.h["state"][0]
        if value is None:
            rec[side * 4:side * 4 + 4] = 0.0
        else:
            rec[side * 4:side * 4 + 3] = np.asarray(value, np.float64).reshape(3)
            rec[side * 4 + 3] = 1.0

    prev_left_fit = property(lambda self: self._get_fit(0), lambda self, v: self._set_fit(0, v))
    prev_right_fit = property(lambda self: self._get_fit(1), lambda self, v: self._set_fit(1, v))

    def _prepare(self, h, w):
        if self._shape == (h, w):
            return
        d = self._dev
        nbytes = int(d.lib.av_lane_workspace_bytes(1, h, w, self.MAX_SEGMENTS))
        self._ws = d.empty(nbytes, torch.uint8)
        nat.check(d.lib.av_lane_workspace_init(d.ctx.handle, d.stream, 1, h, w, self.MAX_SEGMENTS, nat.ptr(self._ws)))
        self._stage = Packed(d, [("frame", np.uint8, (1, h, w, 3))], mapped=False)      # pinned staging for the frame upload
        self._roi = None
        if self.roi_vertices is not None:
            self._roi = d.upload(_roi_rows(self.roi_vertices, h, w), np.int32)
        self._shape = (h, w)

    def _view(self, what, dtype, shape):
        h, w = self._shape
        off, nb = C.c_size_t(), C.c_size_t()
        nat.check(self._dev.lib.av_lane_workspace_view(what, 1, h, w, self.MAX_SEGMENTS, C.byref(off), C.byref(nb)))
        raw = self._ws[off.value:off.value + nb.value].cpu().numpy()
        return raw.view(dtype).reshape(shape)

    def _run(self, frame, stages=0):
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("frame must be an HxWx3 uint8 BGR image, got shape %s" % (frame.shape,))
        h, w = frame.shape[:2]
        self._prepare(h, w)
        d = self._dev
        self._stage.upload_from("frame", frame)          # host copy and DMA pipelined in four pieces
        io = self._io
        cfg = nat.LaneCfg(50, 50, 150, self.MAX_SEGMENTS, float(self.smoothing_factor))
        nat.check(d.lib.av_lane_detect(d.ctx.handle, d.stream, C.byref(cfg), 1, h, w, self._stage.ptr("frame"),
                                       nat.ptr(self._roi), nat.ptr(self._ws), io.ptr("state"), io.ptr("poly"),
                                       io.ptr("pts"), io.ptr("info"), io.ptr("conf"), stages))
        io.download()

    def detect(self, frame: np.ndarray) -> Tuple[Optional[LaneLine], Optional[LaneLine]]:
        self._run(frame)
        h = self._io.h
        info, poly, pts, conf = h["info"][0], h["poly"][0], h["pts"][0], h["conf"][0]
        out = []
        for side, name in ((0, "left"), (1, "right")):
            if info[side]:
                out.append(LaneLine(points=pts[side].copy(), side=name, confidence=float(conf[side]),
                                    polynomial=poly[side].copy()))
            else:
                out.append(None)
        return out[0], out[1]

    def draw_lanes(self, frame: np.ndarray, left_lane: Optional[LaneLine], right_lane: Optional[LaneLine],
                   fill_lane: bool = True) -> np.ndarray:
        """Lane area + the two fitted lines (lane_detector.py:220-251), drawn by the device rasteriser."""
        from ..visualization._prims import PrimList, paint
        pl = PrimList()
        if fill_lane and left_lane is not None and right_lane is not None:
            pl.blend_polygon(np.vstack([left_lane.points, right_lane.points[::-1]]), (0, 255, 100))
        if left_lane is not None:
            pl.polylines(left_lane.points, False, (255, 0, 0), 3)
        if right_lane is not None:
            pl.polylines(right_lane.points, False, (0, 0, 255), 3)
        return paint(frame, pl, self._dev.index)

    def get_lane_center_offset(self, frame_width: int, left_lane: Optional[LaneLine],
                               right_lane: Optional[LaneLine]) -> Optional[float]:
        if left_lane is None or right_lane is None:
            return None
        lane_center = (left_lane.points[-1, 0] + right_lane.points[-1, 0]) / 2
        return frame_width / 2 - lane_center

    def reset(self):
        self._io.h["state"][:] = 0.0
