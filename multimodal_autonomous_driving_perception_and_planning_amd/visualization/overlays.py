"""OverlayRenderer -- drop-in surface of src/visualization/overlays.py, drawn by the device rasteriser.

Panels (0.7/0.3 blended rectangles), text lines, the lane-offset gauge and the side-by-side view keep the reference's
layout (overlays.py:26-210); pixel rules and the font are this project's own (parity unpinned, OpenCV absent).
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev
from ._prims import PrimList, paint


class OverlayRenderer:
    def __init__(self, device: int = 0):
        self.font = 0                     # cv2.FONT_HERSHEY_SIMPLEX
        self.font_scale = 0.5
        self.font_thickness = 1
        self._device = device

    def info_panel_prims(self, pl, vehicle_state=None, fps: float = 0.0, frame_num: int = 0):
        pl.blend_rectangle((10, 10), (250, 150), (0, 0, 0))
        lines = ["Frame: %d" % frame_num, "FPS: %.1f" % fps]
        if vehicle_state:
            lines += ["Speed: %.1f km/h" % (vehicle_state.speed * 3.6), "Heading: %.1f deg" % np.degrees(vehicle_state.heading),
                      "Accel: %.2f m/s2" % vehicle_state.acceleration, "Pos: (%.1f, %.1f)" % (vehicle_state.x, vehicle_state.y)]
        y = 30
        for line in lines:
            pl.put_text(line, (20, y), self.font_scale, (255, 255, 255), self.font_thickness)
            y += 20

    def draw_info_panel(self, frame, vehicle_state=None, fps: float = 0.0, frame_num: int = 0) -> np.ndarray:
        pl = PrimList()
        self.info_panel_prims(pl, vehicle_state, fps, frame_num)
        return paint(frame, pl, self._device)

    def detection_summary_prims(self, pl, w, h, detections, position="top_right"):
        counts = {}
        for det in detections:
            counts[det.class_name] = counts.get(det.class_name, 0) + 1
        x0, y0 = (w - 150, 10) if position == "top_right" else (10, h - 100)
        pl.blend_rectangle((x0, y0), (x0 + 140, y0 + 20 + len(counts) * 18), (0, 0, 0))
        pl.put_text("Detections:", (x0 + 5, y0 + 15), 0.4, (255, 255, 255), 1)
        y = y0 + 35
        for name, n in counts.items():
            pl.put_text("  %s: %d" % (name, n), (x0 + 5, y), 0.35, (200, 200, 200), 1)
            y += 18

    def draw_detection_summary(self, frame, detections: List, position: str = "top_right") -> np.ndarray:
        pl = PrimList()
        self.detection_summary_prims(pl, frame.shape[1], frame.shape[0], detections, position)
        return paint(frame, pl, self._device)

    def draw_lane_offset_indicator(self, frame, offset: Optional[float]) -> np.ndarray:
        h, w = frame.shape[:2]
        iw, ih = 200, 30
        x0, y0 = (w - iw) // 2, h - 50
        pl = PrimList()
        pl.rectangle((x0, y0), (x0 + iw, y0 + ih), (50, 50, 50), -1)
        pl.rectangle((x0, y0), (x0 + iw, y0 + ih), (100, 100, 100), 1)
        cx = x0 + iw // 2
        pl.line((cx, y0), (cx, y0 + ih), (255, 255, 255), 1)
        if offset is not None:
            off = int(np.clip(offset, -100, 100))
            color = (0, 255, 0) if abs(offset) < 20 else ((0, 255, 255) if abs(offset) < 50 else (0, 0, 255))
            pl.circle((cx + off, y0 + ih // 2), 8, color, -1)
            pl.put_text("Offset: %.0fpx" % offset, (x0 + 5, y0 - 5), 0.4, (255, 255, 255), 1)
        return paint(frame, pl, self._device)

    def draw_tracking_stats(self, frame, tracks: List, position: str = "bottom_left") -> np.ndarray:
        h, w = frame.shape[:2]
        x0, y0 = (10, h - 80) if position == "bottom_left" else (w - 150, h - 80)
        pl = PrimList()
        pl.blend_rectangle((x0, y0), (x0 + 140, y0 + 70), (0, 0, 0))
        avg_age = np.mean([t.age for t in tracks]) if tracks else 0
        pl.put_text("Tracking Stats:", (x0 + 5, y0 + 15), 0.4, (255, 255, 255), 1)
        pl.put_text("  Active: %d" % len(tracks), (x0 + 5, y0 + 35), 0.35, (200, 200, 200), 1)
        pl.put_text("  Avg Age: %.0f frames" % avg_age, (x0 + 5, y0 + 55), 0.35, (200, 200, 200), 1)
        return paint(frame, pl, self._device)

    def create_side_by_side(self, frame1, frame2, labels: Tuple[str, str] = ("Camera", "BEV")) -> np.ndarray:
        """Both pictures at the taller one's height (bilinear resize on the device), side by side, labelled."""
        d = Dev(self._device)
        f1, f2 = np.ascontiguousarray(frame1, np.uint8), np.ascontiguousarray(frame2, np.uint8)
        (h1, w1), (h2, w2) = f1.shape[:2], f2.shape[:2]
        th = max(h1, h2)
        nw1 = w1 if h1 == th else int(w1 * (th / h1))
        nw2 = w2 if h2 == th else int(w2 * (th / h2))
        out = d.empty((th, nw1 + nw2, 3), torch.uint8)
        for src, (sh, sw), nw, x0 in ((f1, (h1, w1), nw1, 0), (f2, (h2, w2), nw2, nw1)):
            if sh == th:
                out[:, x0:x0 + nw].copy_(torch.as_tensor(src))
            else:
                s = d.upload(src, np.uint8)
                nat.check(d.lib.av_resize_into(d.ctx.handle, d.stream, nat.ptr(s), sh, sw, nat.ptr(out), th, nw, nw1 + nw2, x0))
        pl = PrimList()
        pl.put_text(labels[0], (10, 25), 0.6, (255, 255, 255), 2)
        pl.put_text(labels[1], (nw1 + 10, 25), 0.6, (255, 255, 255), 2)
        return paint(out.cpu().numpy(), pl, self._device)
