"""The CPU oracle against golden vectors captured from the real reference (tests/golden/make_golden.py)."""
import numpy as np

from oracle.detector_ref import detection_table, centers
from oracle.planner_ref import PlannerRef, trajectory_length
from oracle.tracker_ref import TrackerRef, iou_matrix, greedy_match


def test_detector_matches_reference_bitwise(golden):
    g = golden("detections")
    for tag, (h, w) in {"720": (720, 1280), "480": (480, 640)}.items():
        n, box, cls, conf = detection_table(1, 1100, h, w)
        assert np.array_equal(n, g["n_" + tag])
        assert np.array_equal(box, g["box_" + tag])
        assert np.array_equal(cls, g["cls_" + tag])
        assert np.array_equal(conf, g["conf_" + tag])          # float64, bit-for-bit
    # seed wraps with frame_count % 1000 but positions depend on frame_count itself
    assert not np.array_equal(g["box_720"][0], g["box_720"][1000])
    assert np.array_equal(g["n_720"][0], g["n_720"][1000])


def _replay(g, **kw):
    trk = TrackerRef(**kw)
    nfr = len(g["in_n"])
    for f in range(nfr):
        n = int(g["in_n"][f])
        r = trk.update(n, g["in_box"][f], g["in_cls"][f], g["in_conf"][f])
        t = trk.table(64)
        assert t["n"] == g["n_live"][f], f
        assert trk.next_id == g["next_id"][f]
        assert np.array_equal(t["ids"], g["ids"][f]), f
        assert np.array_equal(t["box"], g["box"][f]), f
        assert np.array_equal(t["cls"], g["cls"][f]), f
        assert np.array_equal(t["conf"], g["conf"][f]), f
        assert np.array_equal(t["ahm"], g["ahm"][f]), f
        assert np.array_equal(r["det2trk"], g["det2trk"][f][:n]), f
        nm = int(g["n_match"][f])
        assert [tuple(p) for p in g["match_pairs"][f][:nm]] == r["pairs"], f
        conf_ids = trk.confirmed_ids()
        assert conf_ids == list(g["conf_ids"][f][:g["n_conf"][f]]), f
        key = "traj_%d" % (f + 1)
        if key in g.files:
            for k, row in enumerate(trk.rows):
                tl, vl = g["traj_len_%d" % (f + 1)][k], g["vel_len_%d" % (f + 1)][k]
                assert len(row["traj"]) == tl and len(row["vel"]) == vl
                assert np.array_equal(np.array(row["traj"]), g[key][k, :tl])
                if vl:
                    assert np.array_equal(np.array(row["vel"]), g["vel_%d" % (f + 1)][k, :vl])
    return trk


def test_tracker_sim720_matches_reference(golden):
    trk = _replay(golden("tracker_sim720"))
    assert trk.next_id > 100          # many births/deaths exercised


def test_tracker_ties_and_short_lifetimes(golden):
    _replay(golden("tracker_ties"), iou_threshold=0.5, max_age=2, min_hits=1, trajectory_length=5)


def test_iou_and_greedy_edge_cases():
    assert iou_matrix([[0, 0, 10, 10]], [[10, 0, 20, 10]])[0, 0] == 0.0          # touching -> 0
    assert iou_matrix([[0, 0, 0, 0]], [[0, 0, 0, 0]])[0, 0] == 0.0               # degenerate
    assert iou_matrix([[0, 0, 10, 10]], [[0, 0, 10, 10]])[0, 0] == 1.0
    m = np.array([[0.5, 0.5], [0.5, 0.5]])
    assert greedy_match(m, 0.3) == [(0, 0), (1, 1)]                               # first in row-major order
    assert greedy_match(np.array([[0.3]]), 0.3) == [(0, 0)]                       # max < thr is strict
    assert greedy_match(np.zeros((0, 3)), 0.3) == []


def _check_plan(p, state, wp=None, cost=None, order=None, obstacles=None):
    r = p.plan(state, obstacles)
    if wp is not None:
        assert np.array_equal(r["wp"], wp)
    assert np.array_equal(r["cost"], cost)
    assert np.array_equal(r["order"], order)
    return r


def test_planner_matches_reference_bitwise(golden):
    g = golden("planner")
    p = PlannerRef()
    for s, st in enumerate(g["states"]):
        r = _check_plan(p, st, wp=g["wp_first8"][s] if s < 8 else None, cost=g["cost"][s], order=g["order"][s])
        assert np.array_equal(r["types"], g["types"][s])
        assert np.allclose(r["wp"].sum(axis=(0, 1)), g["wp_checksum"][s], rtol=1e-13)
        ln = np.array([trajectory_length(w) for w in r["wp"]])
        assert np.array_equal(ln, g["length"][s])
        assert np.array_equal(r["wp"][:, -1, 4] - r["wp"][:, 0, 4], g["duration"][s])
    # exact +/- ties are resolved by generation order
    c0 = g["cost"][0]
    assert c0[0] == c0[18] and list(g["order"][0]).index(0) < list(g["order"][0]).index(18)


def test_planner_ref_path_and_obstacles(golden):
    g = golden("planner")
    obs = [tuple(o) for o in g["obstacles"]]
    for tag, use_ref, use_obs in (("ref", True, False), ("obs", False, True), ("refobs", True, True)):
        p = PlannerRef()
        if use_ref:
            p.set_reference_path(g["ref_path"])
        for s in range(12):
            _check_plan(p, g["states"][s], cost=g["cost_" + tag][s], order=g["order_" + tag][s],
                        obstacles=obs if use_obs else None)


def test_planner_non_default_construction(golden):
    g = golden("planner")
    p = PlannerRef(planning_horizon=3.0, dt=0.2, num_samples=5)
    assert p.n == 16
    for s in range(6):
        _check_plan(p, g["states"][s], wp=g["alt_wp"][s], cost=g["alt_cost"][s], order=g["alt_order"][s])


def test_detection_centers():
    b = np.array([[0, 519, 104, 597]])
    assert centers(b).tolist() == [[52.0, 558.0]]


def test_maneuver_oracle_matches_reference(golden):
    """ManeuverDetector through the real reference module (415 frames, every enum value): bit-for-bit."""
    from oracle import maneuver_ref as M
    g = golden("maneuver")
    idx, val = M.run(g["states"], g["lane_offset"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(val, g["val"])
    assert list(g["lateral_names"]) == list(M.LATERAL) and list(g["longitudinal_names"]) == list(M.LONGITUDINAL)
    assert list(g["turning_names"]) == list(M.TURNING)
    assert set(np.unique(idx[:, 0])) == {0, 1, 2, 3} and set(np.unique(idx[:, 1])) == {0, 1, 2, 3, 4}
    assert set(np.unique(idx[:, 2])) == {0, 1, 2, 3, 4, 5}


def _check_interaction_frame(per, summ, g, f, n):
    if n:
        assert summ["counts"] == tuple(g["counts"][f]), f
    for k in range(n):
        r = per[k]
        if r is None:
            assert g["type"][f, k] == -1, (f, k)
            continue
        assert (r["type"], r["risk"]) == (g["type"][f, k], g["risk"][f, k]), (f, k)
        assert (r["conf"], r["dist"], r["rel"]) == (g["conf"][f, k], g["dist"][f, k], g["rel"][f, k]), (f, k)
        assert (r["ttc"] is None and np.isnan(g["ttc"][f, k])) or r["ttc"] == g["ttc"][f, k], (f, k)
    assert (summ["n_inter"], summ["primary"], summ["overall"]) == (g["n_inter"][f], g["primary"][f], g["overall"][f]), f
    assert summ["closest"] == g["closest"][f] and summ["ts"] == g["ts"][f], f
    assert (summ["min_ttc"] is None and np.isnan(g["min_ttc"][f])) or summ["min_ttc"] == g["min_ttc"][f], f
    assert summ["order"] == list(g["order"][f][:summ["n_inter"]]), f


def test_interaction_oracle_matches_reference(golden):
    """InteractionDetector through the real reference module: handcrafted track lists, and the reference chain
    detector -> tracker -> InteractionDetector replayed with the detector / tracker oracles.  Bit-for-bit, including
    the order the reference's string-keyed sort leaves the interactions in."""
    from oracle import interaction_ref as I
    from oracle.detector_ref import simulated_detections
    from oracle.tracker_ref import TrackerRef
    g = golden("interaction_synth")
    names = list(g["class_names"])
    ref = I.InteractionRef()
    for f in range(len(g["n"])):
        n = int(g["n"][f])
        tr = [dict(id=int(g["ids"][f, k]), kind=I.kind_of(names[g["cls"][f, k]]), bbox=tuple(int(v) for v in g["box"][f, k]),
                   vel=(tuple(g["vel"][f, k]) if g["has_vel"][f, k] else None), conf=g["tconf"][f, k]) for k in range(n)]
        per, summ = ref.detect(tr, g["speed"][f] if g["has_state"][f] else 10.0)
        _check_interaction_frame(per, summ, g, f, n)
    assert {1, 4, 7, 8, 9} <= set(np.unique(g["type"])) and {0, 1, 3} <= set(np.unique(g["overall"]))
    g = golden("interaction")
    det_names = ["car", "truck", "pedestrian", "cyclist", "motorcycle", "bus", "traffic_light", "stop_sign"]
    trk, ref = TrackerRef(), I.InteractionRef()
    for f in range(len(g["speed"])):
        n, box, cls, conf = simulated_detections(f + 1, 720, 1280)
        trk.update(n, box, cls, conf)
        tr = [dict(id=r["id"], kind=I.kind_of(det_names[r["cls"]]), bbox=r["bbox"], vel=(r["vel"][-1] if r["vel"] else None),
                   conf=r["conf"]) for r in trk.rows if r["hits"] >= 3]
        assert [t["id"] for t in tr] == list(g["ids"][f][:g["n_tracks"][f]]), f
        per, summ = ref.detect(tr, g["speed"][f] if f != 77 else 10.0, (720, 1280))
        _check_interaction_frame(per, summ, g, f, len(tr))
    assert {1, 4, 6, 7, 8} <= set(np.unique(g["type"]))
