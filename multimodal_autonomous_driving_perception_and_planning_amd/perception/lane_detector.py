"""LaneDetector -- drop-in surface of src/perception/lane_detector.py (device path in lane.hip)."""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class LaneLine:
    """A fitted lane line (lane_detector.py:13-19)."""
    points: np.ndarray
    side: str
    confidence: float
    polynomial: Optional[np.ndarray] = None


class LaneDetector:
    def __init__(self, roi_vertices: Optional[np.ndarray] = None, device: int = 0):
        self.roi_vertices = roi_vertices
        self.prev_left_fit = None
        self.prev_right_fit = None
        self.smoothing_factor = 0.7

    def detect(self, frame: np.ndarray) -> Tuple[Optional[LaneLine], Optional[LaneLine]]:
        raise NotImplementedError("lane path not built yet")

    def get_lane_center_offset(self, frame_width: int, left_lane: Optional[LaneLine],
                               right_lane: Optional[LaneLine]) -> Optional[float]:
        if left_lane is None or right_lane is None:
            return None
        lane_center = (left_lane.points[-1, 0] + right_lane.points[-1, 0]) / 2
        return frame_width / 2 - lane_center

    def reset(self):
        self.prev_left_fit = None
        self.prev_right_fit = None
