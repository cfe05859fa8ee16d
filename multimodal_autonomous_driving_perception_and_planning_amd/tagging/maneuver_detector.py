"""ManeuverDetector -- drop-in surface of src/tagging/maneuver_detector.py over libavhot.so.

The tags of a frame are computed on the device (av_maneuver_detect) from the current and the 14 previous
vehicle states; `detect_batch` tags whole windows of Kalman output without leaving HBM.  The host keeps the
reference's two 30-deep deques only for get_maneuver_summary().
"""
import ctypes as C
from collections import deque
from dataclasses import dataclass
from enum import Enum
from typing import Dict, List

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev


class LateralManeuver(Enum):
    LANE_KEEPING = "lane_keeping"
    LANE_CHANGE_LEFT = "lane_change_left"
    LANE_CHANGE_RIGHT = "lane_change_right"
    SWERVING = "swerving"


class LongitudinalManeuver(Enum):
    CRUISING = "cruising"
    ACCELERATING = "accelerating"
    BRAKING = "braking"
    HARD_BRAKING = "hard_braking"
    STOPPED = "stopped"


class TurningManeuver(Enum):
    STRAIGHT = "straight"
    TURNING_LEFT = "turning_left"
    TURNING_RIGHT = "turning_right"
    U_TURN = "u_turn"
    CURVING_LEFT = "curving_left"
    CURVING_RIGHT = "curving_right"


_LAT, _LON, _TRN = list(LateralManeuver), list(LongitudinalManeuver), list(TurningManeuver)


@dataclass
class ManeuverTags:
    """Container for maneuver detection results (maneuver_detector.py:44-77)."""
    lateral: LateralManeuver = LateralManeuver.LANE_KEEPING
    lateral_confidence: float = 0.0
    longitudinal: LongitudinalManeuver = LongitudinalManeuver.CRUISING
    longitudinal_confidence: float = 0.0
    turning: TurningManeuver = TurningManeuver.STRAIGHT
    turning_confidence: float = 0.0
    speed_kmh: float = 0.0
    acceleration: float = 0.0
    yaw_rate_deg: float = 0.0
    timestamp: float = 0.0

    def to_dict(self) -> Dict:
        return {
            'lateral': self.lateral.value, 'lateral_confidence': self.lateral_confidence,
            'longitudinal': self.longitudinal.value, 'longitudinal_confidence': self.longitudinal_confidence,
            'turning': self.turning.value, 'turning_confidence': self.turning_confidence,
            'speed_kmh': self.speed_kmh, 'acceleration': self.acceleration, 'yaw_rate_deg': self.yaw_rate_deg,
            'timestamp': self.timestamp,
        }

    def get_tags_list(self) -> List[str]:
        return [self.lateral.value, self.longitudinal.value, self.turning.value]


def _row_to_tags(r) -> ManeuverTags:
    return ManeuverTags(_LAT[int(r["lateral"])], float(r["lateral_confidence"]), _LON[int(r["longitudinal"])],
                        float(r["longitudinal_confidence"]), _TRN[int(r["turning"])], float(r["turning_confidence"]),
                        float(r["speed_kmh"]), float(r["acceleration"]), float(r["yaw_rate_deg"]), float(r["timestamp"]))


class ManeuverDetector:
    LANE_CHANGE_YAW_THRESHOLD = 5.0
    LANE_CHANGE_LATERAL_THRESHOLD = 0.5
    TURN_YAW_RATE_THRESHOLD = 15.0
    HARD_BRAKE_THRESHOLD = -3.0
    BRAKE_THRESHOLD = -1.0
    ACCEL_THRESHOLD = 1.0
    STOPPED_SPEED_THRESHOLD = 0.5

    def __init__(self, history_length: int = 30, device: int = 0):
        if history_length < 15:
            raise ValueError("the device path needs the reference's look-back of 15 states (history_length >= 15)")
        self.history_length = history_length
        self.state_history: deque = deque(maxlen=history_length)
        self.position_history: deque = deque(maxlen=history_length)
        self.frame_count = 0
        self._dev = Dev(device)
        d = self._dev
        self._state = d.zeros(nat.MANEUVER_STATE_DOUBLES, torch.float64)
        self._vs = d.zeros((1, 1, nat.VSTATE_DOUBLES), torch.float64)
        self._off = d.zeros((1, 1), torch.float64)
        self._out = d.zeros(nat.MANEUVER_ROW_BYTES, torch.uint8)

    def detect(self, vehicle_state, lane_offset: float = None) -> ManeuverTags:
        if vehicle_state is None:                                  # :124-125: default tags, nothing recorded
            t = ManeuverTags()
            t.timestamp = self.frame_count / 30.0
            return t
        g = lambda k: float(getattr(vehicle_state, k, 0.0))        # noqa: E731
        speed, heading, acc, yaw, x, y = g("speed"), g("heading"), g("acceleration"), g("yaw_rate"), g("x"), g("y")
        self.state_history.append({'speed': speed, 'heading': heading, 'acceleration': acc, 'yaw_rate': yaw, 'x': x, 'y': y})
        self.position_history.append((x, y))
        d = self._dev
        row = np.zeros(nat.VSTATE_DOUBLES)
        row[0], row[1], row[4], row[5], row[6], row[7] = x, y, heading, speed, acc, yaw
        self._vs.copy_(torch.from_numpy(row).view(1, 1, -1))
        self._off.fill_(float("nan") if lane_offset is None else float(lane_offset))
        nat.check(d.lib.av_maneuver_detect(d.ctx.handle, d.stream, 1, 1, nat.ptr(self._vs), nat.ptr(self._off),
                                           nat.ptr(self._state), nat.ptr(self._out)))
        self.frame_count += 1
        return _row_to_tags(self._out.cpu().numpy().view(nat.MANEUVER_ROW_FIELDS)[0])

    def detect_batch(self, vstate: "torch.Tensor", lane_offset: "torch.Tensor" = None) -> np.ndarray:
        """vstate: device float64 [W, 12] (one stream of av_kf_step output) -> structured array [W] of rows
        (nat.MANEUVER_ROW_FIELDS).  Advances the same state as W detect() calls would."""
        d = self._dev
        W = int(vstate.shape[0])
        out = d.zeros(W * nat.MANEUVER_ROW_BYTES, torch.uint8)
        nat.check(d.lib.av_maneuver_detect(d.ctx.handle, d.stream, 1, W, nat.ptr(vstate), nat.ptr(lane_offset),
                                           nat.ptr(self._state), nat.ptr(out)))
        self.frame_count += W
        return out.cpu().numpy().view(nat.MANEUVER_ROW_FIELDS)

    def get_maneuver_summary(self) -> Dict:
        if len(self.state_history) < 5:
            return {}
        recent = list(self.state_history)[-30:]
        return {
            'avg_speed_kmh': np.mean([s['speed'] for s in recent]) * 3.6,
            'max_speed_kmh': np.max([s['speed'] for s in recent]) * 3.6,
            'min_speed_kmh': np.min([s['speed'] for s in recent]) * 3.6,
            'avg_acceleration': np.mean([s['acceleration'] for s in recent]),
            'max_acceleration': np.max([s['acceleration'] for s in recent]),
            'min_acceleration': np.min([s['acceleration'] for s in recent]),
            'total_distance': self._calculate_distance(),
        }

    def _calculate_distance(self) -> float:
        if len(self.position_history) < 2:
            return 0.0
        p = list(self.position_history)
        total = 0.0
        for i in range(1, len(p)):
            dx, dy = p[i][0] - p[i - 1][0], p[i][1] - p[i - 1][1]
            total += np.sqrt(dx * dx + dy * dy)
        return total

    def reset(self):
        self.state_history.clear()
        self.position_history.clear()
        self.frame_count = 0
        d = self._dev
        nat.check(d.lib.av_maneuver_reset(d.ctx.handle, d.stream, 1, nat.ptr(self._state)))
