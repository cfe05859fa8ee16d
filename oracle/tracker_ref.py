"""Oracle K1-K3: IoU tracker (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Table-based restatement of MultiObjectTracker (src/tracking/multi_object_tracker.py).
The reference keeps a dict keyed by monotonically increasing ids; insertion
order == ascending id, deletions keep order, so a row table kept in id order
with births appended reproduces every "dict order" the reference exposes
(:128 track_ids, :161 unmatched, :236 confirmed).

  _compute_iou  :84-105   exact integer areas, ONE float64 divide
  _associate    :113-164  greedy arg-max, ties -> first in row-major order (:150),
                          stop when max < iou_threshold (:147)
  update        :166-241  match / miss / birth / death / confirmed

Pinned by tests/golden/tracker_sim720.npz and tracker_ties.npz.
"""
import numpy as np


def iou_matrix(tb, db):
    """tb int[T,4], db int[D,4] -> float64[T,D]  (multi_object_tracker.py:84-105, :128-134)."""
    tb = np.asarray(tb, np.int64).reshape(-1, 4)
    db = np.asarray(db, np.int64).reshape(-1, 4)
    xi1 = np.maximum(tb[:, None, 0], db[None, :, 0])
    yi1 = np.maximum(tb[:, None, 1], db[None, :, 1])
    xi2 = np.minimum(tb[:, None, 2], db[None, :, 2])
    yi2 = np.minimum(tb[:, None, 3], db[None, :, 3])
    empty = (xi2 <= xi1) | (yi2 <= yi1)
    inter = (xi2 - xi1) * (yi2 - yi1)
    a1 = (tb[:, 2] - tb[:, 0]) * (tb[:, 3] - tb[:, 1])
    a2 = (db[:, 2] - db[:, 0]) * (db[:, 3] - db[:, 1])
    union = a1[:, None] + a2[None, :] - inter
    ok = (~empty) & (union > 0)
    out = np.zeros(inter.shape, np.float64)
    np.divide(inter, union, out=out, where=ok)
    return out


def greedy_match(iou, thr):
    """-> list of (row, col) in pick order (multi_object_tracker.py:136-159)."""
    m = iou.copy()
    pairs = []
    while m.size:
        if m.max() < thr:
            break
        r, c = np.unravel_index(m.argmax(), m.shape)
        pairs.append((int(r), int(c)))
        m[r, :] = -1
        m[:, c] = -1
    return pairs


class TrackerRef:
    """State is a list of row dicts in ascending-id order."""

    def __init__(self, iou_threshold=0.3, max_age=30, min_hits=3, trajectory_length=50):
        self.iou_threshold = iou_threshold
        self.max_age = max_age
        self.min_hits = min_hits
        self.trajectory_length = trajectory_length
        self.reset()

    def reset(self):
        self.rows = []
        self.next_id = 1
        self.frame_count = 0

    def update(self, n, box, cls, conf):
        """box int[n,4]; returns dict(det2trk, pairs) and mutates the table."""
        self.frame_count += 1
        n = int(n)
        box = np.asarray(box)[:n]
        rows = self.rows
        pairs = []
        if n and rows:
            iou = iou_matrix([r["bbox"] for r in rows], box)
            pairs = greedy_match(iou, self.iou_threshold)
        used_r = {r for r, _ in pairs}
        used_c = {c for _, c in pairs}
        det2trk = np.full(n, -1, np.int64)
        L = self.trajectory_length
        for r, c in pairs:                                    # :182-205
            row = rows[r]
            b = tuple(int(v) for v in box[c])
            ox = (row["bbox"][0] + row["bbox"][2]) / 2
            oy = (row["bbox"][1] + row["bbox"][3]) / 2
            nx = (b[0] + b[2]) / 2
            ny = (b[1] + b[3]) / 2
            row["bbox"] = b
            row["conf"] = float(conf[c])
            row["age"] += 1
            row["hits"] += 1
            row["misses"] = 0
            row["traj"].append((nx, ny))
            row["vel"].append((nx - ox, ny - oy))
            if len(row["traj"]) > L:
                row["traj"] = row["traj"][-L:]
                row["vel"] = row["vel"][-L:]
            det2trk[c] = row["id"]
        for r, row in enumerate(rows):                        # :208-211
            if r not in used_r:
                row["age"] += 1
                row["misses"] += 1
        for c in range(n):                                    # :214-225
            if c in used_c:
                continue
            b = tuple(int(v) for v in box[c])
            rows.append(dict(id=self.next_id, bbox=b, cls=int(cls[c]), conf=float(conf[c]),
                             age=0, hits=1, misses=0,
                             traj=[((b[0] + b[2]) / 2, (b[1] + b[3]) / 2)], vel=[]))
            det2trk[c] = self.next_id
            self.next_id += 1
        self.rows = [r for r in rows if not (r["misses"] > self.max_age)]   # :228-233
        return dict(det2trk=det2trk, pairs=[(rows[r]["id"], c) for r, c in pairs])

    def confirmed_ids(self):
        return [r["id"] for r in self.rows if r["hits"] >= self.min_hits]   # :236-239

    def table(self, tcap):
        """Snapshot as fixed-capacity arrays (same layout the HIP path writes per frame)."""
        T = len(self.rows)
        out = dict(n=T, ids=np.full(tcap, -1, np.int32), box=np.zeros((tcap, 4), np.int32),
                   cls=np.zeros(tcap, np.int32), conf=np.zeros(tcap), ahm=np.zeros((tcap, 3), np.int32))
        for k, r in enumerate(self.rows):
            out["ids"][k] = r["id"]
            out["box"][k] = r["bbox"]
            out["cls"][k] = r["cls"]
            out["conf"][k] = r["conf"]
            out["ahm"][k] = (r["age"], r["hits"], r["misses"])
        return out
