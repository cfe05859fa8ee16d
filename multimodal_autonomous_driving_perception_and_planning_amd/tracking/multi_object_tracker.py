"""MultiObjectTracker -- drop-in surface of src/tracking/multi_object_tracker.py over libavhot.so.

The track table (rows + per-track history rings) lives in HBM and is advanced by
av_tracker_update; this class mirrors it into persistent `Track` objects so callers keep the
reference's object identity semantics (update() returns live objects that are mutated in place).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _native as nat
from .._dev import Dev, Packed

_ROW = np.dtype(nat.TRACK_ROW_FIELDS)


@dataclass
class Track:
    """A tracked object with history (multi_object_tracker.py:14-47)."""
    track_id: int
    bbox: Tuple[int, int, int, int]
    class_id: int
    class_name: str
    confidence: float
    age: int = 0
    hits: int = 1
    misses: int = 0
    trajectory: List[Tuple[float, float]] = field(default_factory=list)
    velocities: List[Tuple[float, float]] = field(default_factory=list)

    @property
    def center(self) -> Tuple[float, float]:
        x1, y1, x2, y2 = self.bbox
        return ((x1 + x2) / 2, (y1 + y2) / 2)

    @property
    def velocity(self) -> Optional[Tuple[float, float]]:
        return self.velocities[-1] if len(self.velocities) > 0 else None

    def predict_next_position(self) -> Tuple[float, float]:
        cx, cy = self.center
        if self.velocity:
            vx, vy = self.velocity
            return (cx + vx, cy + vy)
        return (cx, cy)


class MultiObjectTracker:
    def __init__(self, iou_threshold: float = 0.3, max_age: int = 30, min_hits: int = 3,
                 trajectory_length: int = 50, device: int = 0, capacity: int = 64):
        self.iou_threshold = iou_threshold
        self.max_age = max_age
        self.min_hits = min_hits
        self.trajectory_length = trajectory_length
        self.tracks: Dict[int, Track] = {}
        self.next_id = 1
        self.frame_count = 0
        self._dev = Dev(device)
        self._names: Dict[int, str] = {}
        self._alloc(capacity, 16)
        self._reset_device()

    # ---- device plumbing --------------------------------------------------------------------------
    def _cfg(self):
        return nat.TrackerCfg(float(self.iou_threshold), int(self.max_age), int(self.min_hits),
                              int(self.trajectory_length))

    def _alloc(self, tcap, dcap):
        d = self._dev
        self._tcap, self._dcap, self._L = tcap, dcap, int(self.trajectory_length)
        self._bytes = int(d.lib.av_tracker_state_bytes(tcap, self._L))
        self._state = d.zeros((1, self._bytes), torch.uint8)
        # the frame's detections go up in one copy; the table after the frame (rows carry bbox, counters, history length
        # and the last velocity) comes back in one -- the 100-KB state with the history rings stays on the device
        self._io = Packed(d, [("dn", np.int32, (1, 1)), ("dbox", np.int32, (1, 1, dcap, 4)), ("dcls", np.int32, (1, 1, dcap)),
                              ("dconf", np.float64, (1, 1, dcap)), ("snap_n", np.int32, (1, 1)),
                              ("snap", np.uint8, (1, 1, tcap, nat.TRACK_ROW_BYTES)), ("d2t", np.int32, (1, 1, dcap))])

    def _reset_device(self):
        d = self._dev
        nat.check(d.lib.av_tracker_reset(d.ctx.handle, d.stream, 1, self._tcap, self._L, nat.ptr(self._state)))

    def _host_state(self):
        raw = self._state.cpu().numpy()[0]
        hdr = raw[:nat.TRACKER_HDR_BYTES].view(np.int32)
        ro = nat.TRACKER_HDR_BYTES
        rows = raw[ro:ro + self._tcap * 64].view(_ROW)
        hist = raw[ro + self._tcap * 64:].view(np.float64).reshape(self._tcap, self._L, 4)
        return hdr, rows, hist

    def _grow(self, need_rows, need_dets):
        """Re-allocate with a larger capacity, carrying the table and every history ring over."""
        tcap = self._tcap
        while tcap < need_rows:
            tcap *= 2
        dcap = self._dcap
        while dcap < need_dets:
            dcap *= 2
        if tcap > 1024 or dcap > 64:
            raise RuntimeError("track table capacity exceeded (%d rows, %d detections per frame)" % (need_rows, need_dets))
        hdr, rows, hist = self._host_state()
        old_tcap = self._tcap
        self._alloc(tcap, dcap)
        raw = np.zeros(self._bytes, np.uint8)
        raw[:64] = hdr.view(np.uint8)
        ro = nat.TRACKER_HDR_BYTES
        raw[ro:ro + old_tcap * 64] = rows.view(np.uint8)
        h = raw[ro + tcap * 64:].view(np.float64).reshape(tcap, self._L, 4)
        h[:old_tcap] = hist
        self._state.copy_(torch.as_tensor(raw).view(1, -1))

    # ---- reference surface ----------------------------------------------------------------------------
    def _compute_iou(self, bbox1: Tuple, bbox2: Tuple) -> float:
        """IoU of two boxes (kept for API compatibility; the batched kernel does not call it)."""
        xi1, yi1 = max(bbox1[0], bbox2[0]), max(bbox1[1], bbox2[1])
        xi2, yi2 = min(bbox1[2], bbox2[2]), min(bbox1[3], bbox2[3])
        if xi2 <= xi1 or yi2 <= yi1:
            return 0.0
        inter = (xi2 - xi1) * (yi2 - yi1)
        union = (bbox1[2] - bbox1[0]) * (bbox1[3] - bbox1[1]) + (bbox2[2] - bbox2[0]) * (bbox2[3] - bbox2[1]) - inter
        return inter / union if union > 0 else 0.0

    def update(self, detections: List) -> List[Track]:
        d = self._dev
        if int(self.trajectory_length) != self._L:
            raise RuntimeError("trajectory_length cannot change after construction (history rings are sized by it)")
        n = len(detections)
        if len(self.tracks) + n > self._tcap or n > self._dcap:
            self._grow(len(self.tracks) + n, n)
        io = self._io
        box, cls, conf = io.h["dbox"][0, 0], io.h["dcls"][0, 0], io.h["dconf"][0, 0]
        names = self._names
        for j, det in enumerate(detections):
            box[j] = det.bbox
            cls[j] = det.class_id
            conf[j] = det.confidence
            if det.class_id not in names:
                names[int(det.class_id)] = det.class_name
        box[n:] = 0
        io.h["dn"][0, 0] = n
        io.upload(upto="dconf")
        cfg = self._cfg()
        nat.check(d.lib.av_tracker_update(d.ctx.handle, d.stream, C.byref(cfg), 1, 1, self._dcap, io.ptr("dn"), io.ptr("dbox"),
                                          io.ptr("dcls"), io.ptr("dconf"), self._tcap, nat.ptr(self._state), io.ptr("snap"),
                                          io.ptr("snap_n"), io.ptr("d2t")))
        io.download(first="snap_n")
        self._mirror()
        mh = self.min_hits
        return [t for t in self.tracks.values() if t.hits >= mh]

    def _mirror(self):
        """Bring the Python Track objects in line with the device table (the table is the truth).  One frame at a
        time a track gains at most one history entry -- its new centre and, unless it was just born, the centre
        difference the row carries as its last velocity -- so the rings are only fetched when that does not hold."""
        io = self._io
        n = int(io.h["snap_n"][0, 0])
        if n > self._tcap:
            raise RuntimeError("track table overflow on device (%d rows, capacity %d)" % (n, self._tcap))
        rows = io.h["snap"][0, 0].view(_ROW).reshape(self._tcap)[:n]
        self.frame_count += 1
        L = self._L
        ids, x1, y1, x2, y2 = rows["id"].tolist(), rows["x1"].tolist(), rows["y1"].tolist(), rows["x2"].tolist(), rows["y2"].tolist()
        cls, age, hits, misses = rows["cls"].tolist(), rows["age"].tolist(), rows["hits"].tolist(), rows["misses"].tolist()
        hls, conf, vx, vy = rows["hist_len"].tolist(), rows["conf"].tolist(), rows["vx"].tolist(), rows["vy"].tolist()
        old, new, hist = self.tracks, {}, None
        for k in range(n):
            tid = ids[k]
            t = old.get(tid)
            hl = hls[k]
            if t is None:
                cid = cls[k]
                t = Track(track_id=tid, bbox=(0, 0, 0, 0), class_id=cid, class_name=self._names.get(cid, str(cid)), confidence=0.0)
                t._hist_len = 0
            t.bbox = (x1[k], y1[k], x2[k], y2[k])
            t.confidence = conf[k]
            t.age, t.hits, t.misses = age[k], hits[k], misses[k]
            gap = hl - t._hist_len
            if gap == 1:                               # matched or born in this frame
                t.trajectory.append(((x1[k] + x2[k]) / 2, (y1[k] + y2[k]) / 2))
                if hl > 1:
                    t.velocities.append((vx[k], vy[k]))
                if len(t.trajectory) > L:
                    del t.trajectory[:-L]
                    del t.velocities[:-L]
                t._hist_len = hl
            elif gap != 0:                             # not reachable one frame at a time: rebuild from the device rings
                if hist is None:
                    hist = self._host_state()[2]
                slot = int(rows["slot"][k])
                for e in range(max(t._hist_len, hl - L), hl):
                    cx, cy, ex, ey = hist[slot, e % L]
                    t.trajectory.append((float(cx), float(cy)))
                    if e > 0:
                        t.velocities.append((float(ex), float(ey)))
                if len(t.trajectory) > L:
                    t.trajectory = t.trajectory[-L:]
                    t.velocities = t.velocities[-L:]
                t._hist_len = hl
            new[tid] = t
        self.tracks = new
        if n:
            self.next_id = max(self.next_id, ids[-1] + 1)       # rows are in ascending id order; ids are never re-used

    def get_all_trajectories(self) -> Dict[int, List[Tuple[float, float]]]:
        return {tid: t.trajectory.copy() for tid, t in self.tracks.items() if t.hits >= self.min_hits}

    def draw_tracks(self, frame: np.ndarray, tracks: List[Track], draw_trajectories: bool = True,
                    draw_ids: bool = True, draw_velocities: bool = False) -> np.ndarray:
        """Boxes, ids, trails, velocity arrows (multi_object_tracker.py:251-313), drawn by the device rasteriser."""
        from ..visualization._prims import PrimList, paint
        palette = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255), (0, 255, 255), (128, 0, 255), (255, 128, 0)]
        pl = PrimList()
        for t in tracks:
            col = palette[t.track_id % len(palette)]
            x1, y1, x2, y2 = t.bbox
            pl.rectangle((x1, y1), (x2, y2), col, 2)
            if draw_ids:
                pl.put_text("ID:%d %s" % (t.track_id, t.class_name), (x1, y1 - 10), 0.5, col, 2)
            if draw_trajectories and len(t.trajectory) > 1:
                pts = np.array(t.trajectory, dtype=np.int32)
                for i in range(1, len(pts)):
                    pl.line(tuple(pts[i - 1]), tuple(pts[i]), col, max(1, int(3 * i / len(pts))))
            if draw_velocities and t.velocity:
                cx, cy = int(t.center[0]), int(t.center[1])
                pl.arrowed_line((cx, cy), (int(cx + t.velocity[0] * 5), int(cy + t.velocity[1] * 5)), (0, 255, 255), 2, tip_length=0.3)
        return paint(frame, pl, self._dev.index)

    def reset(self):
        self.tracks.clear()
        self.next_id = 1
        self.frame_count = 0
        self._reset_device()
