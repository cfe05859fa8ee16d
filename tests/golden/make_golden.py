#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference code.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (it never travels to the GPU box; only the .npz files written
here do).  Importable reference stages:

  * src.perception.detector.ObjectDetector._detect_simulated   (detector.py:125-169)
  * src.tracking.MultiObjectTracker                            (multi_object_tracker.py:50-319)
  * src.planning.MotionPlanner                                 (motion_planner.py:57-374)
  * src/tagging/maneuver_detector.py ManeuverDetector          (loaded from its file; NumPy only)
  * src/tagging/interaction_detector.py InteractionDetector    (loaded from its file; NumPy only)

`cv2` is absent here; detector.py imports it at module level but the simulated
path never touches it, so an empty module object is registered under that name
before import.  filterpy / cv2 / ultralytics paths are NOT importable: no
fixtures exist for the Kalman, lane and YOLO stages ("parity unpinned").

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("AV_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not found at %s" % REF)
    sys.path.insert(0, REF)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import importlib

    det = importlib.import_module("src.perception.detector")
    trk = importlib.import_module("src.tracking.multi_object_tracker")
    pln = importlib.import_module("src.planning.motion_planner")
    return det, trk, pln


DMAX = 8     # simulated detector emits 3..7
TMAX = 64    # observed <= 48 live tracks


def golden_detections(det_mod):
    """frames 1..1100 at two shapes (covers the frame_count % 1000 wrap)."""
    out = {}
    for tag, (h, w) in {"720": (720, 1280), "480": (480, 640)}.items():
        d = det_mod.ObjectDetector(mode="simulated")
        frame = np.zeros((h, w, 3), np.uint8)
        nfr = 1100
        n = np.zeros(nfr, np.int32)
        box = np.zeros((nfr, DMAX, 4), np.int32)
        cls = np.zeros((nfr, DMAX), np.int32)
        conf = np.zeros((nfr, DMAX), np.float64)
        for f in range(nfr):
            dets = d.detect(frame)
            n[f] = len(dets)
            for j, x in enumerate(dets):
                box[f, j] = x.bbox
                cls[f, j] = x.class_id
                conf[f, j] = x.confidence
                assert x.center == ((x.bbox[0] + x.bbox[2]) / 2, (x.bbox[1] + x.bbox[3]) / 2)
                assert x.class_name == det_mod.ObjectDetector.CLASSES[int(x.class_id)]
        out["n_" + tag] = n
        out["box_" + tag] = box
        out["cls_" + tag] = cls
        out["conf_" + tag] = conf
    np.savez_compressed(os.path.join(OUT, "detections.npz"), **out)
    return out


class _Det:
    """Minimal stand-in for Detection when feeding hand-made boxes to the reference tracker."""

    def __init__(self, det_mod, bbox, cls, conf):
        self.d = det_mod.Detection(bbox=tuple(int(v) for v in bbox), class_id=int(cls),
                                   class_name=det_mod.ObjectDetector.CLASSES[int(cls)],
                                   confidence=float(conf))


def _run_tracker(det_mod, trk_mod, n, box, cls, conf, hist_frames, **kw):
    """Drive the reference tracker; record the full table after every update."""
    nfr = len(n)
    trk = trk_mod.MultiObjectTracker(**kw)
    assoc = {}
    orig = trk._associate

    def spy(dets):
        r = orig(dets)
        assoc["m"] = r
        return r

    trk._associate = spy
    n_live = np.zeros(nfr, np.int32)
    n_conf = np.zeros(nfr, np.int32)
    next_id = np.zeros(nfr, np.int32)
    ids = np.full((nfr, TMAX), -1, np.int32)
    tb = np.zeros((nfr, TMAX, 4), np.int32)
    tcls = np.zeros((nfr, TMAX), np.int32)
    tconf = np.zeros((nfr, TMAX), np.float64)
    ahm = np.zeros((nfr, TMAX, 3), np.int32)          # age, hits, misses
    conf_ids = np.full((nfr, TMAX), -1, np.int32)      # returned (confirmed) ids, in order
    det2trk = np.full((nfr, DMAX), -1, np.int32)       # detection j -> track id (matched or born)
    n_match = np.zeros(nfr, np.int32)
    match_pairs = np.full((nfr, DMAX, 2), -1, np.int32)  # (track_id, det_idx) in pick order
    hist = {}
    for f in range(nfr):
        dets = []
        for j in range(int(n[f])):
            dets.append(_Det(det_mod, box[f, j], cls[f, j], conf[f, j]).d)
        nid_before = trk.next_id
        res = trk.update(dets)
        matched, unm_t, unm_d = assoc["m"]
        n_match[f] = len(matched)
        for k, (tid, dj) in enumerate(matched):
            match_pairs[f, k] = (tid, dj)
            det2trk[f, dj] = tid
        for k, dj in enumerate(unm_d):
            det2trk[f, dj] = nid_before + k
        live = list(trk.tracks.values())
        assert len(live) <= TMAX
        n_live[f] = len(live)
        next_id[f] = trk.next_id
        for r, t in enumerate(live):
            ids[f, r] = t.track_id
            tb[f, r] = t.bbox
            tcls[f, r] = t.class_id
            tconf[f, r] = t.confidence
            ahm[f, r] = (t.age, t.hits, t.misses)
        n_conf[f] = len(res)
        for r, t in enumerate(res):
            conf_ids[f, r] = t.track_id
        if (f + 1) in hist_frames:
            L = kw.get("trajectory_length", 50)
            tl = np.zeros(TMAX, np.int32)
            vl = np.zeros(TMAX, np.int32)
            tr = np.zeros((TMAX, L, 2), np.float64)
            ve = np.zeros((TMAX, L, 2), np.float64)
            for r, t in enumerate(live):
                tl[r] = len(t.trajectory)
                vl[r] = len(t.velocities)
                if tl[r]:
                    tr[r, :tl[r]] = np.array(t.trajectory)
                if vl[r]:
                    ve[r, :vl[r]] = np.array(t.velocities)
            hist["traj_len_%d" % (f + 1)] = tl
            hist["vel_len_%d" % (f + 1)] = vl
            hist["traj_%d" % (f + 1)] = tr
            hist["vel_%d" % (f + 1)] = ve
    out = dict(n_live=n_live, n_conf=n_conf, next_id=next_id, ids=ids, box=tb, cls=tcls,
               conf=tconf, ahm=ahm, conf_ids=conf_ids, det2trk=det2trk, n_match=n_match,
               match_pairs=match_pairs, in_n=n, in_box=box, in_cls=cls, in_conf=conf)
    out.update(hist)
    return out


def golden_tracker(det_mod, trk_mod, dets):
    # (1) the demo configuration: simulated detections at 720p, defaults, 300 frames
    nfr = 300
    g = _run_tracker(det_mod, trk_mod, dets["n_720"][:nfr], dets["box_720"][:nfr],
                     dets["cls_720"][:nfr], dets["conf_720"][:nfr],
                     hist_frames={60, 150, 300})
    np.savez_compressed(os.path.join(OUT, "tracker_sim720.npz"), **g)

    # (2) adversarial: exact IoU ties, duplicate boxes, empty frames, short lifetimes
    rng = np.random.RandomState(1234)
    nfr = 160
    n = np.zeros(nfr, np.int32)
    box = np.zeros((nfr, DMAX, 4), np.int32)
    cls = np.zeros((nfr, DMAX), np.int32)
    conf = np.zeros((nfr, DMAX), np.float64)
    base = np.array([[100, 100, 160, 160], [100, 100, 160, 160], [130, 100, 190, 160],
                     [400, 300, 480, 360], [400, 300, 480, 360], [700, 500, 760, 560],
                     [70, 100, 130, 160], [900, 200, 990, 290]], np.int32)
    for f in range(nfr):
        if f % 17 == 5 or 60 <= f < 66:
            n[f] = 0
            continue
        k = int(rng.randint(1, DMAX + 1))
        sel = rng.permutation(DMAX)[:k]
        jit = rng.randint(-2, 3, size=(k, 1)) * (rng.rand(k, 1) < 0.5)
        b = base[sel] + np.concatenate([jit, jit, jit, jit], axis=1)
        n[f] = k
        box[f, :k] = b
        cls[f, :k] = rng.randint(0, 8, size=k)
        conf[f, :k] = rng.uniform(0.1, 1.0, size=k)
    g = _run_tracker(det_mod, trk_mod, n, box, cls, conf, hist_frames={40, 160},
                     iou_threshold=0.5, max_age=2, min_hits=1, trajectory_length=5)
    np.savez_compressed(os.path.join(OUT, "tracker_ties.npz"), **g)


def _traj_arrays(trajs):
    C = len(trajs)
    npts = len(trajs[0].waypoints)
    wp = np.zeros((C, npts, 6), np.float64)
    cost = np.zeros(C, np.float64)
    for c, t in enumerate(trajs):
        cost[c] = t.cost
        for i, w in enumerate(t.waypoints):
            wp[c, i] = (w.x, w.y, w.heading, w.velocity, w.timestamp, w.curvature)
    return wp, cost


TYPE_CODE = {"lane_keep": 0, "lane_change_left": 1, "lane_change_right": 2}


def _plan_case(pln_mod, state, ref=None, obstacles=None, **kw):
    """Returns waypoints in GENERATION order, costs in generation order, order = sorted->generation idx."""
    p = pln_mod.MotionPlanner(**kw)
    if ref is not None:
        p.set_reference_path([tuple(r) for r in ref])
    # capture generation order by wrapping evaluate (called once per candidate, in generation order)
    gen = []
    orig = p.evaluate_trajectory_cost

    def spy(traj, obstacles=None):
        gen.append(traj)
        return orig(traj, obstacles)

    p.evaluate_trajectory_cost = spy
    opt, cands = p.plan(tuple(state), obstacles)
    wp, cost = _traj_arrays(gen)
    pos = {id(t): k for k, t in enumerate(gen)}
    order = np.array([pos[id(t)] for t in cands], np.int32)
    assert cands[0] is opt
    types_ = np.array([TYPE_CODE[t.trajectory_type] for t in gen], np.int32)
    length = np.array([t.length for t in gen])
    duration = np.array([t.duration for t in gen])
    return wp, cost, order, types_, length, duration


def golden_planner(pln_mod):
    rng = np.random.RandomState(7)
    states = [(0.0, 0.0, 0.0, 10.0), (0.0, 0.0, 0.0, 0.0), (12.5, -3.25, 0.3, 9.0),
              (100.0, 50.0, -1.2, 14.0), (-7.0, 2.0, 3.1, 6.5), (1.0, 2.0, np.pi / 2, 11.0),
              (5.0, 0.0, 0.0, 8.0), (0.0, 7.5, -np.pi, 12.0)]
    while len(states) < 32:
        states.append((float(rng.uniform(-200, 200)), float(rng.uniform(-200, 200)),
                       float(rng.uniform(-np.pi, np.pi)), float(rng.uniform(0, 20))))
    states = np.array(states, np.float64)
    S = len(states)
    out = {"states": states}
    wp_all = np.zeros((S, 21, 51, 6))
    cost = np.zeros((S, 21))
    order = np.zeros((S, 21), np.int32)
    types_ = np.zeros((S, 21), np.int32)
    length = np.zeros((S, 21))
    duration = np.zeros((S, 21))
    for s in range(S):
        wp_all[s], cost[s], order[s], types_[s], length[s], duration[s] = _plan_case(pln_mod, states[s])
    out.update(cost=cost, order=order, types=types_, length=length, duration=duration)
    out["wp_first8"] = wp_all[:8]            # full waypoint arrays for 8 states
    out["wp_checksum"] = wp_all.sum(axis=(1, 2))  # [S,6] column sums for the rest

    # reference path + obstacles variants
    ref = np.stack([np.linspace(0, 60, 13), 0.02 * np.linspace(0, 60, 13) ** 1.5], axis=1)
    obstacles = [(20.0, 1.0, 1.5), (35.0, -2.0, 2.0), (8.0, 3.0, 0.8)]
    vs = states[:12]
    for tag, r, o in (("ref", ref, None), ("obs", None, obstacles), ("refobs", ref, obstacles)):
        c = np.zeros((len(vs), 21))
        od = np.zeros((len(vs), 21), np.int32)
        for s in range(len(vs)):
            _, c[s], od[s], _, _, _ = _plan_case(pln_mod, vs[s], ref=r, obstacles=o)
        out["cost_" + tag] = c
        out["order_" + tag] = od
    out["ref_path"] = ref
    out["obstacles"] = np.array(obstacles)

    # non-default construction: horizon 3.0, dt 0.2, 5 lateral samples -> 15 candidates x 16 points
    kw = dict(planning_horizon=3.0, dt=0.2, num_samples=5)
    wps, cs, ods = [], [], []
    for s in range(6):
        w, c, od, _, _, _ = _plan_case(pln_mod, states[s], **kw)
        wps.append(w), cs.append(c), ods.append(od)
    out["alt_wp"] = np.array(wps)
    out["alt_cost"] = np.array(cs)
    out["alt_order"] = np.array(ods)

    # set_reference_path headings (motion_planner.py:93-124)
    p = pln_mod.MotionPlanner()
    p.set_reference_path([tuple(r) for r in ref])
    out["ref_heading"] = np.array([w.heading for w in p.reference_trajectory.waypoints])
    np.savez_compressed(os.path.join(OUT, "planner.npz"), **out)


def _import_tagger(name):
    """The tagging package's __init__ pulls in transformers/VLM code; the two rule-based detectors are plain
    NumPy modules, so they are loaded straight from their files."""
    import importlib.util
    path = os.path.join(REF, "src", "tagging", name + ".py")
    spec = importlib.util.spec_from_file_location("ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def maneuver_inputs():
    """A state sequence that visits every branch of ManeuverDetector (maneuver_detector.py:105-262):
    noisy cruise, steady yaw (lane-change rule), hard/normal braking, stop, acceleration, 90-degree turns,
    curves, a u-turn whose heading change needs the +-360 normalisation, and lane offsets incl. None."""
    rs = np.random.RandomState(11)
    seg = []

    def add(n, speed, acc, yaw, dhead, off):
        seg.append((n, speed, acc, yaw, dhead, off))

    add(40, 10.0, 0.0, 0.0, 0.0, None)                 # cruise, no offset
    add(30, 10.0, 0.2, 0.12, 0.004, 0.3)               # steady left yaw > 5 deg/s
    add(30, 10.0, -0.2, -0.12, -0.004, -0.3)           # steady right yaw
    add(25, 9.0, -1.5, 0.0, 0.0, 0.7)                  # braking, offset left
    add(25, 6.0, -3.6, 0.0, 0.0, -0.7)                 # hard braking, offset right
    add(20, 0.2, 0.0, 0.0, 0.0, None)                  # stopped
    add(25, 4.0, 1.6, 0.0, 0.0, 0.1)                   # accelerating
    add(40, 8.0, 0.0, 0.6, 0.045, None)                # left turn (heading ramps ~0.63 rad per 14 frames -> curving/turning)
    add(40, 8.0, 0.0, -0.6, -0.09, None)               # right turn, faster
    add(30, 5.0, 0.0, 1.2, 0.2, None)                  # u-turn rate (> 120 deg per 14 frames, wraps past 180)
    add(30, 10.0, 0.0, 0.3, 0.0, None)                 # instantaneous yaw-rate fallback (> 15 deg/s), no heading change
    add(40, 10.0, 0.0, 0.0, 0.0, 0.51)                 # offset just over the threshold
    states, offs = [], []
    x = y = head = 0.0
    for n, speed, acc, yaw, dhead, off in seg:
        for _ in range(n):
            head += dhead
            x += speed * np.cos(head) / 30.0
            y += speed * np.sin(head) / 30.0
            states.append((speed + rs.normal(0, 0.02), head, acc + rs.normal(0, 0.05), yaw + rs.normal(0, 0.01), x, y))
            offs.append(np.nan if off is None else off)
    # a noisy stretch: yaw-rate std > 0.1 -> swerving
    for _ in range(40):
        head += rs.normal(0, 0.01)
        states.append((10.0, head, rs.normal(0, 0.3), rs.normal(0, 0.25), x, y))
        offs.append(np.nan)
    return np.array(states, np.float64), np.array(offs, np.float64)


def golden_maneuver():
    mod = _import_tagger("maneuver_detector")
    states, offs = maneuver_inputs()
    det = mod.ManeuverDetector()
    lat = [m for m in mod.LateralManeuver]
    lon = [m for m in mod.LongitudinalManeuver]
    trn = [m for m in mod.TurningManeuver]
    idx = np.zeros((len(states), 3), np.int32)
    val = np.zeros((len(states), 7), np.float64)
    for i, (s, o) in enumerate(zip(states, offs)):
        vs = types.SimpleNamespace(speed=s[0], heading=s[1], acceleration=s[2], yaw_rate=s[3], x=s[4], y=s[5])
        t = det.detect(vs, None if np.isnan(o) else float(o))
        idx[i] = (lat.index(t.lateral), lon.index(t.longitudinal), trn.index(t.turning))
        val[i] = (t.lateral_confidence, t.longitudinal_confidence, t.turning_confidence, t.speed_kmh, t.acceleration,
                  t.yaw_rate_deg, t.timestamp)
    summary = det.get_maneuver_summary()
    np.savez_compressed(os.path.join(OUT, "maneuver.npz"), states=states, lane_offset=offs, idx=idx, val=val,
                        lateral_names=np.array([m.value for m in lat]), longitudinal_names=np.array([m.value for m in lon]),
                        turning_names=np.array([m.value for m in trn]),
                        summary_keys=np.array(sorted(summary)), summary_vals=np.array([summary[k] for k in sorted(summary)]))
    print("maneuver.npz: %d frames; lateral %s longitudinal %s turning %s" % (
        len(states), np.bincount(idx[:, 0], minlength=4), np.bincount(idx[:, 1], minlength=5), np.bincount(idx[:, 2], minlength=6)))


def golden_interaction(det_mod, trk_mod, dets):
    """The reference chain detector -> tracker -> InteractionDetector on the 720p simulated detections, 300 frames,
    ego speed varying around 10 m/s.  Recorded per frame: the summary fields, and per returned (confirmed) track in
    tracker order the Interaction it produced (type -1 = none), plus the order of tags.interactions after the
    reference's sort (agent ids), which decides primary_interaction."""
    mod = _import_tagger("interaction_detector")
    nfr = 300
    n, box, cls, conf = dets["n_720"][:nfr], dets["box_720"][:nfr], dets["cls_720"][:nfr], dets["conf_720"][:nfr]
    trk = trk_mod.MultiObjectTracker()
    itd = mod.InteractionDetector()
    types_ = [t for t in mod.InteractionType]
    risks = [r for r in mod.RiskLevel]
    speeds = 10.0 + 3.0 * np.sin(np.arange(nfr) * 0.05)
    K = TMAX
    out = dict(speed=speeds, n_tracks=np.zeros(nfr, np.int32), ids=np.full((nfr, K), -1, np.int32),
               type=np.full((nfr, K), -1, np.int32), risk=np.zeros((nfr, K), np.int32), conf=np.zeros((nfr, K)),
               dist=np.zeros((nfr, K)), rel=np.zeros((nfr, K)), ttc=np.full((nfr, K), np.nan),
               counts=np.zeros((nfr, 4), np.int32), n_inter=np.zeros(nfr, np.int32), primary=np.full(nfr, -1, np.int32),
               overall=np.zeros(nfr, np.int32), closest=np.zeros(nfr), min_ttc=np.full(nfr, np.nan), ts=np.zeros(nfr),
               order=np.full((nfr, K), -1, np.int32))
    for f in range(nfr):
        d = [_Det(det_mod, box[f, j], cls[f, j], conf[f, j]).d for j in range(int(n[f]))]
        tracks = trk.update(d)
        vs = types.SimpleNamespace(speed=float(speeds[f]), x=0.0, y=0.0) if f != 77 else None     # one frame without state
        # per-track records need the un-sorted association track -> interaction: replay _analyze on a spy
        per = {}
        orig = itd._analyze_interaction

        def spy(track, *a, **k):
            r = orig(track, *a, **k)
            per[track.track_id] = r
            return r

        itd._analyze_interaction = spy
        tags = itd.detect(tracks, vs, frame_shape=(720, 1280))
        itd._analyze_interaction = orig
        out["n_tracks"][f] = len(tracks)
        for k, t in enumerate(tracks):
            out["ids"][f, k] = t.track_id
            r = per.get(t.track_id)
            if r is not None:
                out["type"][f, k] = types_.index(r.type)
                out["risk"][f, k] = risks.index(r.risk_level)
                out["conf"][f, k], out["dist"][f, k], out["rel"][f, k] = r.confidence, r.distance, r.relative_speed
                out["ttc"][f, k] = np.nan if r.time_to_collision is None else r.time_to_collision
        out["counts"][f] = (tags.agent_count, tags.pedestrian_count, tags.cyclist_count, tags.vehicle_count)
        out["n_inter"][f] = len(tags.interactions)
        out["primary"][f] = -1 if tags.primary_interaction is None else types_.index(tags.primary_interaction)
        out["overall"][f] = risks.index(tags.overall_risk)
        out["closest"][f] = tags.closest_agent_distance
        out["min_ttc"][f] = np.nan if tags.min_ttc is None else tags.min_ttc
        out["ts"][f] = tags.timestamp
        for k, it in enumerate(tags.interactions):
            out["order"][f, k] = it.agent_id
    out["type_names"] = np.array([t.value for t in types_])
    out["risk_names"] = np.array([r.value for r in risks])
    np.savez_compressed(os.path.join(OUT, "interaction.npz"), **out)
    tt = out["type"][out["type"] >= 0]
    print("interaction.npz: %d frames; types seen %s; overall risk %s; cut-ins %d" % (
        nfr, dict(zip(*np.unique(tt, return_counts=True))), np.bincount(out["overall"], minlength=4), int((tt == 4).sum())))


def golden_interaction_synth():
    """Handcrafted track lists through the real InteractionDetector.detect (default frame_shape 480x640): persistent
    ids so histories build up, boxes big and low enough for NEAR_MISS, degenerate heights, velocity None, class names
    outside the detector's list, frames with no tracks, tracks that disappear and slots that are re-used."""
    mod = _import_tagger("interaction_detector")
    itd = mod.InteractionDetector()
    types_ = [t for t in mod.InteractionType]
    risks = [r for r in mod.RiskLevel]
    names = ["car", "truck", "pedestrian", "cyclist", "motorcycle", "bus", "traffic_light", "stop_sign", "bicycle", "person"]
    rs = np.random.RandomState(5)
    nfr, K = 160, 12
    live = {}                       # slot -> dict(id, cls, x, y, w, h, dx)
    next_id = 1
    out = dict(n=np.zeros(nfr, np.int32), ids=np.full((nfr, K), -1, np.int32), slot=np.full((nfr, K), -1, np.int32),
               box=np.zeros((nfr, K, 4), np.int32), cls=np.zeros((nfr, K), np.int32), tconf=np.zeros((nfr, K)),
               vel=np.zeros((nfr, K, 2)), has_vel=np.zeros((nfr, K), np.int32), speed=np.zeros(nfr), has_state=np.ones(nfr, np.int32),
               type=np.full((nfr, K), -1, np.int32), risk=np.zeros((nfr, K), np.int32), conf=np.zeros((nfr, K)),
               dist=np.zeros((nfr, K)), rel=np.zeros((nfr, K)), ttc=np.full((nfr, K), np.nan),
               counts=np.zeros((nfr, 4), np.int32), n_inter=np.zeros(nfr, np.int32), primary=np.full(nfr, -1, np.int32),
               overall=np.zeros(nfr, np.int32), closest=np.zeros(nfr), min_ttc=np.full(nfr, np.nan), ts=np.zeros(nfr),
               order=np.full((nfr, K), -1, np.int32))
    for f in range(nfr):
        # births / deaths
        for s in list(live):
            if rs.rand() < 0.04:
                del live[s]
        while len(live) < (0 if 60 <= f < 64 else rs.randint(3, K)):
            free = [s for s in range(K) if s not in live]
            if not free:
                break
            s = free[0]
            live[s] = dict(id=next_id, cls=int(rs.randint(0, len(names))), x=float(rs.randint(0, 560)), y=float(rs.randint(150, 420)),
                           w=int(rs.randint(20, 120)), h=int(rs.choice([0, -3, 20, 40, 80, 120, 200])), dx=float(rs.randint(-6, 7)), age=0)
            next_id += 1
        if 60 <= f < 64:
            live.clear()
        tracks = []
        order_slots = sorted(live, key=lambda s: live[s]["id"])          # tracker order: ascending id
        for s in order_slots:
            t = live[s]
            vy = float(rs.randint(-8, 9)) / 2.0
            vel = None if t["age"] == 0 or rs.rand() < 0.1 else (t["dx"], vy)
            t["x"] += t["dx"]
            t["age"] += 1
            x1, y1 = int(t["x"]), int(t["y"])
            bbox = (x1, y1, x1 + t["w"], y1 + t["h"])
            if 100 <= f < 135 and itd._estimate_distance(bbox, (480, 640)) < 3.0:
                continue                      # slow stretch without near-misses: overall risk comes from the rules
            tracks.append(types.SimpleNamespace(track_id=t["id"], class_name=names[t["cls"]], bbox=bbox, velocity=vel,
                                                confidence=float(rs.uniform(0.3, 1.0)), _slot=s, _cls=t["cls"]))
        speed = float(rs.uniform(0.0, 20.0)) if not 100 <= f < 135 else float(rs.uniform(0.0, 0.3))   # slow stretch: few TTCs
        vs = None if f % 37 == 36 else types.SimpleNamespace(speed=speed, x=0.0, y=0.0)
        out["speed"][f], out["has_state"][f] = speed, 0 if vs is None else 1
        per = {}
        orig = itd._analyze_interaction

        def spy(track, *a, **k):
            r = orig(track, *a, **k)
            per[track.track_id] = r
            return r

        itd._analyze_interaction = spy
        tags = itd.detect(tracks, vs)
        itd._analyze_interaction = orig
        out["n"][f] = len(tracks)
        for k, t in enumerate(tracks):
            out["ids"][f, k], out["slot"][f, k], out["box"][f, k], out["cls"][f, k] = t.track_id, t._slot, t.bbox, t._cls
            out["tconf"][f, k] = t.confidence
            if t.velocity is not None:
                out["vel"][f, k], out["has_vel"][f, k] = t.velocity, 1
            r = per.get(t.track_id)
            if r is not None:
                out["type"][f, k], out["risk"][f, k] = types_.index(r.type), risks.index(r.risk_level)
                out["conf"][f, k], out["dist"][f, k], out["rel"][f, k] = r.confidence, r.distance, r.relative_speed
                out["ttc"][f, k] = np.nan if r.time_to_collision is None else r.time_to_collision
        out["counts"][f] = (tags.agent_count, tags.pedestrian_count, tags.cyclist_count, tags.vehicle_count)
        out["n_inter"][f] = len(tags.interactions)
        out["primary"][f] = -1 if tags.primary_interaction is None else types_.index(tags.primary_interaction)
        out["overall"][f] = risks.index(tags.overall_risk)
        out["closest"][f] = tags.closest_agent_distance
        out["min_ttc"][f] = np.nan if tags.min_ttc is None else tags.min_ttc
        out["ts"][f] = tags.timestamp
        for k, it in enumerate(tags.interactions):
            out["order"][f, k] = it.agent_id
    out["class_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "interaction_synth.npz"), **out)
    tt = out["type"][out["type"] >= 0]
    print("interaction_synth.npz: %d frames; types seen %s; overall %s; primary %s" % (
        nfr, dict(zip(*np.unique(tt, return_counts=True))), np.bincount(out["overall"], minlength=4),
        dict(zip(*np.unique(out["primary"], return_counts=True)))))


def main():
    det_mod, trk_mod, pln_mod = _import_reference()
    dets = golden_detections(det_mod)
    golden_tracker(det_mod, trk_mod, dets)
    golden_planner(pln_mod)
    golden_maneuver()
    golden_interaction(det_mod, trk_mod, dets)
    golden_interaction_synth()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("%-24s %8d B" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
