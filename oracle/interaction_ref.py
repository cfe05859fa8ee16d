"""Oracle T2: rule-based interaction tags (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restatement of InteractionDetector.detect (src/tagging/interaction_detector.py:132-222) and its helpers:

  _estimate_distance        :224-248  perspective model from the box bottom and height, clamped to [2, 100]
  _estimate_relative_speed  :250-260  ego_speed - vy, 0.0 when the track has no velocity yet
  _calculate_ttc            :262-268  distance / relative_speed when relative_speed > 0.1
  _analyze_interaction      :270-372  near-miss / pedestrian / cyclist / following / cut-in rules, in that order
  _calculate_overall_risk   :374-395
  sort at :214                        key (risk_level.value, -confidence), reverse=True: the risk STRINGS are compared
                                      ('medium' > 'low' > 'high' > 'critical'), then ascending confidence, stable

Types / risks are indices into the reference's Enum definition order (:20-40).  Pinned by
tests/golden/interaction.npz (reference detector -> tracker -> InteractionDetector, 300 frames) and
interaction_synth.npz (handcrafted track lists through the real module).
"""
import numpy as np

TYPES = ("no_interaction", "following_vehicle", "being_followed", "yielding", "vehicle_cut_in", "vehicle_cut_out",
         "pedestrian_crossing", "pedestrian_waiting", "cyclist_nearby", "near_miss", "merging", "passing", "being_passed")
RISKS = ("low", "medium", "high", "critical")
# class-name categories the rules distinguish
KIND_OTHER, KIND_PEDESTRIAN, KIND_CYCLIST, KIND_CAR, KIND_MOTORCYCLE = 0, 1, 2, 3, 4


def kind_of(class_name):
    if class_name == "pedestrian":
        return KIND_PEDESTRIAN
    if class_name in ("cyclist", "bicycle"):
        return KIND_CYCLIST
    if class_name in ("car", "truck", "bus"):
        return KIND_CAR
    if class_name == "motorcycle":
        return KIND_MOTORCYCLE          # counted as a vehicle (:160) but not covered by the vehicle rules (:330)
    return KIND_OTHER


def estimate_distance(bbox, frame_shape):
    h, w = frame_shape
    x1, y1, x2, y2 = bbox
    bh = y2 - y1
    if bh <= 0:
        return 50.0
    yn = y2 / h
    base = 50.0 * (1 - yn) + 5.0
    size = 100.0 / (bh + 10)
    return max(2.0, min(100.0, (base + size) / 2))


class InteractionRef:
    def __init__(self, history_length=30):
        self.history_length = history_length
        self.reset()

    def reset(self):
        self.hist = {}
        self.frame_count = 0

    def detect(self, tracks, ego_speed=10.0, frame_shape=(480, 640)):
        """tracks: list of dict(id, kind, bbox, vel (tuple or None), conf) in tracker order.
        -> (per-track list of None | dict(type, conf, risk, dist, rel, ttc), summary dict)."""
        summ = dict(counts=(0, 0, 0, 0), n_inter=0, primary=-1, overall=0, closest=float("inf"), min_ttc=None,
                    ts=self.frame_count / 30.0, order=[])
        if not tracks:
            self.frame_count += 1
            return [], summ
        ped = sum(t["kind"] == KIND_PEDESTRIAN for t in tracks)
        cyc = sum(t["kind"] == KIND_CYCLIST for t in tracks)
        veh = sum(t["kind"] in (KIND_CAR, KIND_MOTORCYCLE) for t in tracks)
        h, w = frame_shape
        per, inter = [], []
        min_d, min_ttc = float("inf"), float("inf")
        for t in tracks:
            bbox = t["bbox"]
            d = estimate_distance(bbox, frame_shape)
            min_d = min(min_d, d)
            rel = 0.0 if t["vel"] is None else ego_speed - t["vel"][1]
            ttc = d / rel if rel > 0.1 else None
            if ttc is not None and ttc > 0:
                min_ttc = min(min_ttc, ttc)
            hq = self.hist.setdefault(t["id"], [])
            hq.append((bbox[0] + bbox[2]) / 2)
            del hq[:-self.history_length]
            r = self._analyze(t, d, rel, ttc, w, hq)
            per.append(r)
            if r is not None:
                inter.append((t["id"], r))
        if inter:
            # sorted(key=(risk string, -conf), reverse=True), stable
            srt = sorted(inter, key=lambda e: (RISKS[e[1]["risk"]], -e[1]["conf"]), reverse=True)
            summ["order"] = [e[0] for e in srt]
            summ["primary"] = srt[0][1]["type"]
            if min_ttc and min_ttc < 1.5:
                summ["overall"] = 3
            else:
                summ["overall"] = max(e[1]["risk"] for e in inter)
        summ["counts"] = (len(tracks), ped, cyc, veh)
        summ["n_inter"] = len(inter)
        summ["closest"] = min_d if min_d != float("inf") else 0
        summ["min_ttc"] = None if min_ttc == float("inf") else min_ttc
        self.frame_count += 1
        return per, summ

    @staticmethod
    def _analyze(t, d, rel, ttc, w, hq):
        bbox, kind = t["bbox"], t["kind"]
        cx = (bbox[0] + bbox[2]) / 2
        if d < 3.0:
            return dict(type=9, conf=0.9, risk=3, dist=d, rel=rel, ttc=ttc)
        if kind == KIND_PEDESTRIAN and d < 10.0:
            if abs(cx - w / 2) < w / 4:
                return dict(type=6, conf=0.8, risk=2 if d < 8 else 1, dist=d, rel=rel, ttc=ttc)
            return dict(type=7, conf=0.6, risk=0, dist=d, rel=0.0, ttc=None)
        if kind == KIND_CYCLIST and d < 15:
            return dict(type=8, conf=0.7, risk=1 if d < 8 else 0, dist=d, rel=rel, ttc=None)
        if kind == KIND_CAR:
            if w / 4 < cx < 3 * w / 4 and 5.0 < d < 30.0:
                risk = 0
                if d < 10:
                    risk = 1
                if ttc and ttc < 3.0:
                    risk = 2
                return dict(type=1, conf=0.75, risk=risk, dist=d, rel=rel, ttc=ttc)
            if len(hq) >= 10 and abs(hq[-1] - w / 2) < abs(hq[0] - w / 2) and d < 15.0:
                return dict(type=4, conf=0.7, risk=1, dist=d, rel=rel, ttc=None)
        return None
