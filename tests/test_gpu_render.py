"""f-2: device rasteriser, BEV panel and overlays vs oracle/raster_ref.py (parity unpinned against cv2: OpenCV absent)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch


def _random_list(rng, n, w, h, P):
    pl = P.PrimList()
    for _ in range(n):
        t = rng.randint(0, 9)
        col = tuple(int(v) for v in rng.randint(0, 256, 3))
        pt = lambda: (int(rng.randint(-20, w + 20)), int(rng.randint(-20, h + 20)))      # noqa: E731
        if t == 0:
            pl.rectangle(pt(), pt(), col, -1)
        elif t == 1:
            pl.rectangle(pt(), pt(), col, int(rng.randint(1, 4)))
        elif t == 2:
            a = pt()
            pl.line(a, a if rng.rand() < 0.2 else pt(), col, int(rng.randint(1, 6)))
        elif t == 3:
            c = np.array(pt())
            d = rng.randint(5, 60, 2)
            ang = rng.uniform(0, np.pi)
            u, v = np.array([np.cos(ang), np.sin(ang)]) * d[0], np.array([-np.sin(ang), np.cos(ang)]) * d[1]
            pl.fill_convex_quad([c + u + v, c + u - v, c - u - v, c - u + v], col)
        elif t == 4:
            pl.circle(pt(), int(rng.randint(0, 30)), col, -1)
        elif t == 5:
            pl.circle(pt(), int(rng.randint(1, 40)), col, 1)
        elif t == 6:
            pl.put_text("ID:%d Ab/9%%" % rng.randint(0, 999), pt(), float(rng.choice([0.35, 0.4, 0.5, 0.6])), col, 1)
        elif t == 7:
            pl.blend_rectangle(pt(), pt(), col)
        else:
            k = int(rng.randint(3, 9))
            c, r = np.array(pt()), rng.randint(10, 70, k)
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            pl.blend_polygon(np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)], 1).astype(np.int64), col)
            pl.arrowed_line(pt(), pt(), col, 2, tip_length=0.3)
    return pl


def test_rasteriser_matches_oracle_on_random_lists(gpu):
    from multimodal_autonomous_driving_perception_and_planning_amd.visualization import _prims as P
    from oracle import raster_ref as R
    rng = np.random.RandomState(3)
    for (h, w, n) in ((211, 333, 400), (64, 96, 60), (97, 40, 150)):
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        pl = _random_list(rng, n, w, h, P)
        got = P.paint(img, pl)
        want = R.draw(img, pl.array(), pl.vert_array())
        assert got.shape == want.shape and np.array_equal(got, want), (h, w, int((got != want).sum()))
    # more primitives on one tile than the LDS list holds (1024): the flush must keep the painting order
    img = np.zeros((40, 40, 3), np.uint8)
    pl = P.PrimList()
    for k in range(2600):
        pl.rectangle((k % 7, k % 5), (39 - k % 11, 39 - k % 3), (k % 256, (3 * k) % 256, (7 * k) % 256), -1)
        if k % 9 == 0:
            pl.line((0, k % 40), (39, (2 * k) % 40), (255 - k % 256, 0, k % 256), 1 + k % 3)
    assert np.array_equal(P.paint(img, pl), R.draw(img, pl.array(), pl.vert_array()))
    assert np.array_equal(P.paint(img, P.PrimList()), img)                       # an empty list paints nothing


def test_bev_panel_and_class_overlays(gpu, monkeypatch):
    """BEVRenderer.render / OverlayRenderer / the draw_* methods on the objects of a real 40-frame loop: every picture
    equals the oracle's painting of the same primitive list, and the layout the reference defines is there."""
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion, synthetic_frame
    from multimodal_autonomous_driving_perception_and_planning_amd.visualization import _prims as P
    from oracle import raster_ref as R
    from src.perception import LaneDetector, ObjectDetector
    from src.planning import MotionPlanner
    from src.state_estimation import VehicleStateEstimator
    from src.tracking import MultiObjectTracker
    from src.visualization import BEVRenderer, OverlayRenderer
    det, lane, trk, est, pl = ObjectDetector(mode="simulated"), LaneDetector(), MultiObjectTracker(), VehicleStateEstimator(), MotionPlanner()
    ego = generate_ego_motion(40)
    frame = synthetic_frame(720, 1280, 0, 3)
    for i in range(40):
        dets = det.detect(frame)
        tracks = trk.update(dets)
        st = est.step(np.array(ego[i]))
        opt, cands = pl.plan((st.x, st.y, st.heading, st.speed))
    left, right = lane.detect(frame)
    assert tracks and left is not None and right is not None
    painted = []
    real_paint = P.paint

    def spy(img, plist, device=0):
        out = real_paint(img, plist, device)
        painted.append((np.array(img), plist, out))
        return out
    monkeypatch.setattr(P, "paint", spy)
    import multimodal_autonomous_driving_perception_and_planning_amd.visualization.bev_renderer as B
    import multimodal_autonomous_driving_perception_and_planning_amd.visualization.overlays as O
    monkeypatch.setattr(B, "paint", spy)
    monkeypatch.setattr(O, "paint", spy)
    bev = BEVRenderer()
    assert bev.world_to_pixel(0.0, 0.0) == (300, 500) and bev.pixel_to_world(300, 500) == (0.0, 0.0)
    base = bev.create_base_image()
    assert base.shape == (600, 600, 3) and tuple(base[300, 5]) == bev.bg_color and tuple(base[300, 300 - 40]) == bev.road_color
    panel = bev.render(ego_state=st, tracks=tracks, planned_trajectory=opt, candidate_trajectories=cands[:10], show_grid=True)
    scene = bev.scene_prims(st, tracks, opt, cands[:10], True)
    assert np.array_equal(panel, R.draw(base, scene.array(), scene.vert_array()))
    ex, ey = bev.world_to_pixel(st.x, st.y)
    if 0 <= ex < 600 and 0 <= ey < 600:
        near = panel[max(ey - 12, 0):ey + 13, max(ex - 12, 0):ex + 13].reshape(-1, 3)
        assert (near == np.array(bev.ego_color, np.uint8)).all(axis=1).any()        # the ego footprint is on the panel
    assert tuple(panel[15, 17]) == bev.ego_color and tuple(panel[35, 17]) == (0, 255, 0)             # legend swatches
    assert (panel == np.array((0, 255, 0), np.uint8)).all(axis=2).sum() > 200       # the planned path is drawn
    # camera-view overlays in demo.py's order (demo.py:124-137), then the side-by-side view (:149-151)
    ov = OverlayRenderer()
    cam = det.draw_detections(frame, dets)
    cam = lane.draw_lanes(cam, left, right)
    cam = trk.draw_tracks(cam, tracks, draw_velocities=True)
    cam = pl.draw_trajectories(cam, opt, cands)
    cam = ov.draw_info_panel(cam, st, fps=31.5, frame_num=40)
    cam = ov.draw_detection_summary(cam, dets)
    cam = ov.draw_lane_offset_indicator(cam, lane.get_lane_center_offset(1280, left, right))
    cam = ov.draw_tracking_stats(cam, tracks)
    assert cam.shape == frame.shape and not np.array_equal(cam, frame)
    assert len(painted) >= 9
    for img, plist, out in painted:
        assert np.array_equal(out, R.draw(img, plist.array(), plist.vert_array()))
    # info panel: a 0.7/0.3 blend of black over (10,10)-(250,150)
    assert np.array_equal(ov.draw_info_panel(frame)[12:20, 200:240], ((7 * frame[12:20, 200:240].astype(np.uint32) + 5) // 10).astype(np.uint8))
    both = ov.create_side_by_side(cam, panel, ("Camera View", "Bird's Eye View"))
    assert both.shape == (720, 1280 + 720, 3)
    want = np.hstack([cam, R.resize(panel, 720, 720)])
    lab = painted[-1][1]
    assert np.array_equal(both, R.draw(want, lab.array(), lab.vert_array()))


def test_device_built_bev_panels_match_the_class_renderer(gpu):
    """HotLoop.enqueue_bev builds the panel's primitive list on the device from the tables a step left in HBM; the same
    panel through the class API (Track / VehicleState / Trajectory objects made from those tables) must come out the
    same -- exactly, except where the ego footprint's corners round differently (device cos/sin vs NumPy's)."""
    from types import SimpleNamespace
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from multimodal_autonomous_driving_perception_and_planning_amd.planning.motion_planner import Trajectory
    from oracle.harness_ref import ego_motion
    from src.visualization import BEVRenderer
    S, W = 3, 45
    loop = HotLoop(n_streams=S, window=W)
    loop.reset(frame_offsets=[0, 17, 340])
    loop.load_measurements(np.stack([ego_motion(W, seed=s) for s in range(S)]))
    loop.step(sync=True)
    loop.enqueue_bev()
    loop.synchronize()
    got = loop.bev.cpu().numpy()
    r = loop.results()
    rows, n = loop.snapshots()
    hdr, trows, hist = loop.tracker_tables()
    L = loop.tcfg.trajectory_length
    bev = BEVRenderer()
    for s in range(S):
        tracks = []
        for k in range(n[s, W - 1]):
            row = rows[s, W - 1, k]
            if not row["flags"] & 1:
                continue
            hl, slot = int(row["hist_len"]), int(row["slot"])
            traj = [(float(hist[s, slot, e % L, 0]), float(hist[s, slot, e % L, 1])) for e in range(max(0, hl - L), hl)]
            bbox = (int(row["x1"]), int(row["y1"]), int(row["x2"]), int(row["y2"]))
            tracks.append(SimpleNamespace(track_id=int(row["id"]), bbox=bbox, trajectory=traj,
                                          center=((bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2)))
        v = r["vstate"][s, W - 1]
        st = SimpleNamespace(x=v[0], y=v[1], heading=v[4], pos_uncertainty=v[9])
        cands = [Trajectory._from_array(r["wp"][s, W - 1, c], cost=float(r["cost"][s, W - 1, c])) for c in r["order"][s, W - 1]]
        want = bev.render(ego_state=st, tracks=tracks, planned_trajectory=cands[0], candidate_trajectories=cands[:10])
        diff = (got[s] != want).any(axis=2)
        assert len(tracks) > 0 and diff.mean() < 2e-4, (s, float(diff.mean()))
        assert np.array_equal(got[s][:60, :120], want[:60, :120])                   # the legend corner, exactly


def test_video_loader_uncompressed_formats(gpu, tmp_path):
    """f-4: VideoDataLoader on a Y4M (4:2:0) and a .npy file -- the reference's surface (video_loader.py:14-259), frames
    converted / resized on the device and equal to the oracle's conversion; compressed containers are refused loudly."""
    from data.loaders.video_loader import VideoDataLoader
    from data import VideoDataLoader as V2
    from oracle import raster_ref as R
    assert V2 is VideoDataLoader
    rng = np.random.RandomState(9)
    h, w, n = 48, 64, 5
    frames_yuv = [rng.randint(0, 256, size=h * w * 3 // 2).astype(np.uint8) for _ in range(n)]
    p = tmp_path / "clip.y4m"
    with open(p, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C420jpeg\n" % (w, h))
        for fr in frames_yuv:
            f.write(b"FRAME\n" + fr.tobytes())
    ld = VideoDataLoader(str(p))
    assert (len(ld), ld.total_frames, ld.fps, ld.width, ld.height, ld.dt) == (n, n, 25.0, w, h, 1 / 25.0)
    assert abs(ld.duration - n / 25.0) < 1e-12 and ld.get_info()["total_frames"] == n
    want = [R.i420_to_bgr(fr, h, w) for fr in frames_yuv]
    got = list(ld)
    assert len(got) == n and all(np.array_equal(a, b) for a, b in zip(got, want))
    assert np.array_equal(ld.read_frame_at(3), want[3]) and ld.frame_count == 4
    assert ld.read_frame_at(n) is None and ld.read_frame_at(-1) is None
    assert np.array_equal(ld.read_frame(), want[4]) and ld.read_frame() is None            # end of video
    ld.reset()
    assert len(list(ld.generate_video_stream(2))) == 2
    np.random.seed(3)
    ego = ld.generate_ego_motion()
    assert len(ego) == n and len(ego[0]) == 4
    # target_size: resized on the device like the oracle's bilinear rule; a batch stays in HBM
    ld2 = VideoDataLoader(str(p), target_size=(40, 30))
    assert (ld2.width, ld2.height) == (40, 30)
    assert np.array_equal(ld2.read_frame_at(1), R.resize(want[1], 30, 40))
    batch = ld2.read_frames_device(0, n)
    assert batch.is_cuda and tuple(batch.shape) == (n, 30, 40, 3) and np.array_equal(batch[4].cpu().numpy(), R.resize(want[4], 30, 40))
    # .npy container
    arr = rng.randint(0, 256, size=(3, 20, 24, 3)).astype(np.uint8)
    q = tmp_path / "clip.npy"
    np.save(q, arr)
    ld3 = VideoDataLoader(str(q))
    assert len(ld3) == 3 and np.array_equal(ld3.read_frame_at(2), arr[2])
    ld3.release()
    assert ld3.read_frame() is None
    with pytest.raises(FileNotFoundError):
        VideoDataLoader(str(tmp_path / "missing.mp4"))
    (tmp_path / "x.mp4").write_bytes(b"\x00\x00\x00\x18ftypmp42")
    with pytest.raises(ValueError, match="decoder"):
        VideoDataLoader(str(tmp_path / "x.mp4"))
