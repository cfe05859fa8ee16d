"""InteractionDetector -- drop-in surface of src/tagging/interaction_detector.py over libavhot.so.

detect() packs the track list into one table (row k = k-th track, like the tracker's per-frame snapshot) and
runs av_interaction_detect; the batched entry consumes the tracker's snapshot tables directly.  The list of
interactions is put in the reference's order on the host with the same sort key.
"""
import ctypes as C
from collections import deque
from dataclasses import dataclass, field
from enum import Enum
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev


class InteractionType(Enum):
    NONE = "no_interaction"
    FOLLOWING = "following_vehicle"
    BEING_FOLLOWED = "being_followed"
    YIELDING = "yielding"
    VEHICLE_CUT_IN = "vehicle_cut_in"
    VEHICLE_CUT_OUT = "vehicle_cut_out"
    PEDESTRIAN_CROSSING = "pedestrian_crossing"
    PEDESTRIAN_WAITING = "pedestrian_waiting"
    CYCLIST_NEARBY = "cyclist_nearby"
    NEAR_MISS = "near_miss"
    MERGING = "merging"
    PASSING = "passing"
    BEING_PASSED = "being_passed"


class RiskLevel(Enum):
    LOW = "low"
    MEDIUM = "medium"
    HIGH = "high"
    CRITICAL = "critical"


_TYPES, _RISKS = list(InteractionType), list(RiskLevel)


@dataclass
class Interaction:
    """Single interaction event (interaction_detector.py:43-65)."""
    type: InteractionType
    confidence: float
    risk_level: RiskLevel
    agent_id: Optional[int] = None
    agent_class: Optional[str] = None
    distance: float = 0.0
    relative_speed: float = 0.0
    time_to_collision: Optional[float] = None

    def to_dict(self) -> Dict:
        return {'type': self.type.value, 'confidence': self.confidence, 'risk_level': self.risk_level.value,
                'agent_id': self.agent_id, 'agent_class': self.agent_class, 'distance': self.distance,
                'relative_speed': self.relative_speed, 'time_to_collision': self.time_to_collision}


@dataclass
class InteractionTags:
    """Container for interaction detection results (interaction_detector.py:68-106)."""
    interactions: List[Interaction] = field(default_factory=list)
    primary_interaction: Optional[InteractionType] = None
    overall_risk: RiskLevel = RiskLevel.LOW
    agent_count: int = 0
    pedestrian_count: int = 0
    cyclist_count: int = 0
    vehicle_count: int = 0
    closest_agent_distance: float = float('inf')
    min_ttc: Optional[float] = None
    timestamp: float = 0.0

    def to_dict(self) -> Dict:
        return {
            'interactions': [i.to_dict() for i in self.interactions],
            'primary_interaction': self.primary_interaction.value if self.primary_interaction else None,
            'overall_risk': self.overall_risk.value, 'agent_count': self.agent_count,
            'pedestrian_count': self.pedestrian_count, 'cyclist_count': self.cyclist_count,
            'vehicle_count': self.vehicle_count, 'closest_agent_distance': self.closest_agent_distance,
            'min_ttc': self.min_ttc, 'timestamp': self.timestamp,
        }

    def get_tags_list(self) -> List[str]:
        tags = []
        for interaction in self.interactions:
            if interaction.confidence > 0.5:
                tags.append(interaction.type.value)
        if self.overall_risk != RiskLevel.LOW:
            tags.append(f"risk_{self.overall_risk.value}")
        return list(set(tags))


def class_kind(class_name) -> int:
    """Category the rules distinguish (interaction_detector.py:156-161, :301, :324, :338)."""
    if class_name == 'pedestrian':
        return 1
    if class_name in ('cyclist', 'bicycle'):
        return 2
    if class_name in ('car', 'truck', 'bus'):
        return 3
    if class_name == 'motorcycle':
        return 4
    return 0


def interaction_cfg(frame_shape, class_names=None) -> "nat.InteractionCfg":
    """class_names: list of the detector's class names by id (None: rows already carry the category as `cls`)."""
    cfg = nat.InteractionCfg()
    cfg.frame_h, cfg.frame_w = int(frame_shape[0]), int(frame_shape[1])
    for k in range(16):
        cfg.class_kind[k] = (k if k < 5 else 0) if class_names is None else (class_kind(class_names[k]) if k < len(class_names) else 0)
    return cfg


class InteractionDetector:
    FOLLOWING_DISTANCE_MAX = 30.0
    FOLLOWING_DISTANCE_MIN = 5.0
    NEAR_MISS_DISTANCE = 3.0
    PEDESTRIAN_DANGER_DISTANCE = 10.0
    CUT_IN_DISTANCE = 15.0
    TTC_CRITICAL = 1.5
    TTC_WARNING = 3.0
    CAP = 64                                       # tracks per call on the device path

    def __init__(self, history_length: int = 30, device: int = 0):
        if history_length != 30:
            raise ValueError("the device path keeps the reference's default history of 30 centres per track")
        self.history_length = history_length
        self.track_history: Dict[int, deque] = {}
        self.frame_count = 0
        self._dev = Dev(device)
        d = self._dev
        self._state = d.zeros(int(d.lib.av_interaction_state_bytes(self.CAP)), torch.uint8)
        nat.check(d.lib.av_interaction_reset(d.ctx.handle, d.stream, 1, self.CAP, nat.ptr(self._state)))
        self._slot_of: Dict[int, int] = {}          # track id -> history slot (stable while the id keeps appearing)
        self._last_seen: Dict[int, int] = {}
        self._rows = d.zeros(self.CAP * nat.TRACK_ROW_BYTES, torch.uint8)
        self._n = d.zeros(1, torch.int32)
        self._vy = d.zeros(self.CAP, torch.float64)
        self._vs = d.zeros(nat.VSTATE_DOUBLES, torch.float64)
        self._has = d.zeros(1, torch.uint8)
        self._out = d.zeros(self.CAP * nat.INTERACTION_ROW_BYTES, torch.uint8)
        self._summ = d.zeros(nat.INTERACTION_SUMMARY_BYTES, torch.uint8)

    def _slot(self, tid: int) -> int:
        s = self._slot_of.get(tid)
        if s is None:
            used = set(self._slot_of.values())
            free = [k for k in range(self.CAP) if k not in used]
            if free:
                s = free[0]
            else:                                   # evict the id that has been absent longest
                old = min(self._slot_of, key=lambda i: self._last_seen[i])
                s = self._slot_of.pop(old)
                self._last_seen.pop(old)
            self._slot_of[tid] = s
        self._last_seen[tid] = self.frame_count
        return s

    def detect(self, tracks: List, vehicle_state, frame_shape: Tuple[int, int] = (480, 640)) -> InteractionTags:
        tags = InteractionTags()
        tags.timestamp = self.frame_count / 30.0
        if not tracks:
            self.frame_count += 1
            self._empty_frame()
            return tags
        if len(tracks) > self.CAP:
            raise ValueError("InteractionDetector: more than %d tracks in one call" % self.CAP)
        d = self._dev
        rows = np.zeros(self.CAP, nat.TRACK_ROW_FIELDS)
        vy = np.zeros(self.CAP)
        names = []
        for k, t in enumerate(tracks):
            tid = getattr(t, 'track_id', 0)
            name = getattr(t, 'class_name', 'unknown')
            bbox = getattr(t, 'bbox', (0, 0, 0, 0))
            vel = getattr(t, 'velocity', (0, 0))
            names.append(name)
            r = rows[k]
            r["id"], r["x1"], r["y1"], r["x2"], r["y2"] = tid, bbox[0], bbox[1], bbox[2], bbox[3]
            r["cls"], r["flags"], r["slot"] = class_kind(name), 1, self._slot(tid)
            r["hist_len"] = 1 if vel is None else 2
            if vel is not None:
                vy[k] = vel[1]
            if tid not in self.track_history:
                self.track_history[tid] = deque(maxlen=self.history_length)
            self.track_history[tid].append(((bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2))
        self._rows.copy_(torch.from_numpy(rows.view(np.uint8)))
        self._n.fill_(len(tracks))
        self._vy.copy_(torch.from_numpy(vy))
        self._has.fill_(0 if vehicle_state is None else 1)
        if vehicle_state is not None:
            v = np.zeros(nat.VSTATE_DOUBLES)
            v[5] = float(getattr(vehicle_state, 'speed', 10.0))
            self._vs.copy_(torch.from_numpy(v))
        cfg = interaction_cfg(frame_shape)
        nat.check(d.lib.av_interaction_detect(d.ctx.handle, d.stream, C.byref(cfg), 1, 1, self.CAP, nat.ptr(self._rows),
                                              nat.ptr(self._n), nat.ptr(self._vs), nat.ptr(self._has), nat.ptr(self._vy),
                                              nat.ptr(self._state), nat.ptr(self._out), nat.ptr(self._summ)))
        out = self._out.cpu().numpy().view(nat.INTERACTION_ROW_FIELDS)
        sm = self._summ.cpu().numpy().view(nat.INTERACTION_SUMMARY_FIELDS)[0]
        inter = []
        for k in range(len(tracks)):
            o = out[k]
            if o["type"] < 0:
                continue
            ttc = float(o["ttc"])
            inter.append(Interaction(type=_TYPES[int(o["type"])], confidence=float(o["confidence"]),
                                     risk_level=_RISKS[int(o["risk"])], agent_id=int(o["agent_id"]), agent_class=names[k],
                                     distance=float(o["distance"]), relative_speed=float(o["relative_speed"]),
                                     time_to_collision=None if ttc != ttc else ttc))
        tags.agent_count, tags.pedestrian_count = int(sm["agent_count"]), int(sm["pedestrian_count"])
        tags.cyclist_count, tags.vehicle_count = int(sm["cyclist_count"]), int(sm["vehicle_count"])
        tags.closest_agent_distance = float(sm["closest_distance"])
        mt = float(sm["min_ttc"])
        tags.min_ttc = None if mt != mt else mt
        if inter:
            inter.sort(key=lambda x: (x.risk_level.value, -x.confidence), reverse=True)       # :214
            tags.primary_interaction = _TYPES[int(sm["primary_type"])]
            tags.overall_risk = _RISKS[int(sm["overall_risk"])]
        tags.interactions = inter
        self.frame_count += 1
        return tags

    def _empty_frame(self):
        """An empty track list still advances the device frame counter (its timestamps follow it)."""
        d = self._dev
        self._n.fill_(0)
        self._has.fill_(0)
        cfg = interaction_cfg((480, 640))
        nat.check(d.lib.av_interaction_detect(d.ctx.handle, d.stream, C.byref(cfg), 1, 1, self.CAP, nat.ptr(self._rows),
                                              nat.ptr(self._n), None, nat.ptr(self._has), None, nat.ptr(self._state),
                                              nat.ptr(self._out), nat.ptr(self._summ)))

    def get_interaction_summary(self) -> Dict:
        return {'tracked_agents': len(self.track_history), 'frame_count': self.frame_count}

    def reset(self):
        self.track_history.clear()
        self.frame_count = 0
        self._slot_of.clear()
        self._last_seen.clear()
        d = self._dev
        nat.check(d.lib.av_interaction_reset(d.ctx.handle, d.stream, 1, self.CAP, nat.ptr(self._state)))
