"""Oracle T1: rule-based maneuver tags (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restatement of ManeuverDetector.detect (src/tagging/maneuver_detector.py:105-262), the consumer of the
Kalman output that SURVEY.md section 8(f) ranks third.  Per frame, from the last <= 30 vehicle states:

  _detect_lateral_maneuver      :163-199  np.mean / np.std of the last 10 yaw rates, then the lane offset
  _detect_longitudinal_maneuver :201-227  thresholds on speed and acceleration
  _detect_turning_maneuver      :229-270  heading change over the last 15 states, +-360 normalisation

Enum values are indices into the reference's Enum definition order (:18-41).  Pinned by
tests/golden/maneuver.npz (415 frames through the real module, every enum value visited).
"""
import numpy as np

LATERAL = ("lane_keeping", "lane_change_left", "lane_change_right", "swerving")
LONGITUDINAL = ("cruising", "accelerating", "braking", "hard_braking", "stopped")
TURNING = ("straight", "turning_left", "turning_right", "u_turn", "curving_left", "curving_right")


class ManeuverRef:
    def __init__(self, history_length=30):
        self.history_length = history_length
        self.reset()

    def reset(self):
        self.yaw, self.head = [], []
        self.frame_count = 0

    def step(self, speed, heading, acceleration, yaw_rate, lane_offset=None):
        """-> (idx[3] int, val[7] float64): lateral/longitudinal/turning index; their confidences, speed_kmh,
        acceleration, yaw_rate_deg, timestamp."""
        ts = self.frame_count / 30.0
        self.yaw = (self.yaw + [yaw_rate])[-self.history_length:]
        self.head = (self.head + [heading])[-self.history_length:]
        lat, lat_c = self._lateral(lane_offset)
        lon, lon_c = self._longitudinal(speed, acceleration)
        trn, trn_c = self._turning(yaw_rate)
        self.frame_count += 1
        return (lat, lon, trn), (lat_c, lon_c, trn_c, speed * 3.6, acceleration, np.degrees(yaw_rate), ts)

    def _lateral(self, lane_offset):                                     # :163-199
        if len(self.yaw) >= 10:
            recent = self.yaw[-10:]
            avg, std = np.mean(recent), np.std(recent)
            if std > 0.1:
                return 3, min(0.9, std * 5)
            avg_deg = np.degrees(avg)
            if avg_deg > 5.0:
                return 1, min(0.9, abs(avg_deg) / 20.0)
            elif avg_deg < -5.0:
                return 2, min(0.9, abs(avg_deg) / 20.0)
        if lane_offset is not None and abs(lane_offset) > 0.5:
            return (1, 0.6) if lane_offset > 0 else (2, 0.6)
        return 0, 0.8

    @staticmethod
    def _longitudinal(speed, acc):                                       # :201-227
        if speed < 0.5:
            return 4, 0.95
        if acc < -3.0:
            return 3, min(0.95, abs(acc) / 5.0)
        if acc < -1.0:
            return 2, min(0.9, abs(acc) / 3.0)
        if acc > 1.0:
            return 1, min(0.9, acc / 3.0)
        return 0, 0.8

    def _turning(self, yaw_rate):                                        # :229-270
        yaw_deg = np.degrees(yaw_rate)
        if len(self.head) < 15:
            return 0, 0.5
        recent = self.head[-15:]
        hc = np.degrees(recent[-1] - recent[0])
        while hc > 180:
            hc -= 360
        while hc < -180:
            hc += 360
        if abs(hc) > 120:
            return 3, 0.8
        if hc > 60:
            return 1, min(0.9, hc / 90)
        elif hc < -60:
            return 2, min(0.9, abs(hc) / 90)
        if hc > 15:
            return 4, min(0.8, hc / 45)
        elif hc < -15:
            return 5, min(0.8, abs(hc) / 45)
        if abs(yaw_deg) > 15.0:
            return (4, 0.6) if yaw_deg > 0 else (5, 0.6)
        return 0, 0.8


def run(states, lane_offset=None):
    """states float64[n, >=4] (speed, heading, acceleration, yaw_rate, ...); lane_offset float64[n], NaN = None."""
    m = ManeuverRef()
    idx = np.zeros((len(states), 3), np.int32)
    val = np.zeros((len(states), 7), np.float64)
    for i, s in enumerate(states):
        off = None if lane_offset is None or np.isnan(lane_offset[i]) else float(lane_offset[i])
        idx[i], val[i] = m.step(s[0], s[1], s[2], s[3], off)
    return idx, val
