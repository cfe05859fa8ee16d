"""CPU-only checks: the C-ABI library loads and exports what include/avhot.h declares; host-side logic."""
import ctypes as C
import os

import numpy as np
import pytest

from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(nat.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return nat.lib()


def test_library_exports_every_declared_symbol(lib):
    names = nat.declared_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), "libavhot.so lacks %s declared in include/avhot.h" % n
    assert lib.av_version() == 102
    assert {s[0] for s in nat._SIGS + nat._OPTIONAL_SIGS} >= set(names), "ctypes binding missing for a declared symbol"


def test_struct_layouts_match_header(lib):
    assert np.dtype(nat.TRACK_ROW_FIELDS).itemsize == nat.TRACK_ROW_BYTES == 64
    assert C.sizeof(nat.TrackerCfg) == 24 and C.sizeof(nat.KfCfg) == 24 and C.sizeof(nat.PlannerCfg) == 56
    assert lib.av_tracker_state_bytes(64, 50) == 64 + 64 * 64 + 64 * 50 * 32
    assert lib.av_tracker_state_bytes(0, 50) == 0


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = C.c_int(-1)
    assert lib.av_device_count(C.byref(n)) == 0 and n.value == 0
    h = C.c_void_p()
    rc = lib.av_ctx_create(0, C.byref(h))
    assert rc == -3 and b"no HIP device" in lib.av_last_error_string()
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        from src.planning import MotionPlanner
        MotionPlanner()
    with pytest.raises(RuntimeError):
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        HotLoop()


def test_argument_validation_without_gpu(lib):
    assert lib.av_planner_configure(None, None) == -1
    assert lib.av_ctx_destroy(None) == 0
    assert lib.av_graph_launch(None, 0, None) == -1


def test_dataclass_surfaces():
    from src.perception import ObjectDetector
    from src.planning import Trajectory, Waypoint
    from src.tracking import Track
    from multimodal_autonomous_driving_perception_and_planning_amd.perception import Detection
    d = Detection(bbox=(0, 519, 104, 597), class_id=0, class_name="car", confidence=0.5)
    assert d.center == (52.0, 558.0)
    t = Track(track_id=1, bbox=(0, 0, 10, 20), class_id=0, class_name="car", confidence=0.9)
    assert t.center == (5.0, 10.0) and t.velocity is None and t.predict_next_position() == (5.0, 10.0)
    t.velocities.append((1.0, -2.0))
    assert t.predict_next_position() == (6.0, 8.0)
    a = Trajectory(waypoints=[Waypoint(0, 0, 0, 1, 0.0), Waypoint(3, 4, 0, 1, 0.5)])
    b = Trajectory(waypoints=[Waypoint(0, 0, 0, 1, 0.0), Waypoint(3, 4, 0, 1, 0.5)])
    assert a == b and a.length == 5.0 and a.duration == 0.5 and a.get_positions().shape == (2, 2)
    b.cost = 1.0
    assert a != b
    assert Trajectory(waypoints=[]).length == 0.0 and Trajectory(waypoints=[]).duration == 0.0
    assert ObjectDetector.CLASSES[7] == "stop_sign" and len(ObjectDetector.CLASS_COLORS) == 8


def test_shard_streams_partition():
    from multimodal_autonomous_driving_perception_and_planning_amd.distributed import shard_streams
    for total, world in ((512, 8), (10, 4), (3, 8), (64, 1)):
        spans = [shard_streams(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_streams(512, 8, 3) == (192, 256)


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, tcap = 3, 64
        rows = np.zeros((S, tcap), np.dtype(nat.TRACK_ROW_FIELDS))
        counts = np.zeros(S, np.int32)
        for s in range(S):
            n = 1 + ((rank * S + s) % 5)
            counts[s] = n
            rows["id"][s, :n] = 1000 * (rank * S + s) + np.arange(n)
            rows["conf"][s, :n] = 0.5 + rank
            rows["x2"][s, :n] = 7 * s + rank
        msg = D.pack_tables(torch.as_tensor(rows.view(np.uint8).reshape(S, tcap, 64)), torch.as_tensor(counts))
        out = D.all_gather_tables(msg, world)
        r2, c2 = D.unpack_tables(out, tcap)
        ok = r2.shape == (world * S, tcap)
        for g in range(world * S):
            n = 1 + (g % 5)
            ok &= int(c2[g]) == n and list(r2["id"][g, :n]) == list(1000 * g + np.arange(n))
            ok &= float(r2["conf"][g, 0]) == 0.5 + g // S and int(r2["x2"][g, 0]) == 7 * (g % S) + g // S
        lo, hi = D.shard_streams(world * S, world, rank)
        ok &= (lo, hi) == (rank * S, rank * S + S)
        # window 1 with the one-launch step: begin_step() hands the send buffer to the loop BEFORE the step, the step writes the
        # wire tables itself, exchange() only gathers -- per-frame tables, both buffers re-used, previous gather still intact
        S, W, tcap = 3, 1, 64
        loop = _FakeLoop(S, W, tcap, fused_step=True)
        x = D.TrackTableExchange(loop, world, rank, per_frame=True)
        ok &= x.prepacked and x.bytes_per_step == S * (16 + 32 * tcap)
        prev = None
        for k in range(5):
            x.begin_step()
            ok &= loop.wire.data_ptr() == x.send[k & 1].data_ptr() and loop._wire_ids == (rank * S, 0)
            loop.step_fused(rank, k)
            buf = x.exchange()
            hdr, rows = x.latest()
            ok &= hdr.shape == (world * S, 1)
            for g in range(world * S):
                m = 1 + ((g + 0 + 3 * k) % 7)
                ok &= int(hdr["n_rows"][g, 0]) == m and int(hdr["stream"][g, 0]) == g and int(hdr["frame"][g, 0]) == k + 1
                ok &= list(rows[g, 0]["id"][:m]) == list(100000 * k + 1000 * g + np.arange(m)) and not rows[g, 0]["id"][m:].any()
            if prev is not None:
                h2, _ = D.unpack_wire(prev, tcap)
                ok &= int(h2["frame"][0, 0]) == k          # the other receive buffer still holds step k - 1 (frame index k)
            prev = buf
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, tcap, K = 3, 64, 3
        loop = _FakeLoop(S, 1, tcap, fused_step=True)
        x = D.TrackTableExchange(loop, world, rank, per_frame=True, bucket=K)
        ok = x.prepacked and x.bucket == K and x.bytes_per_gather == K * S * (16 + 32 * tcap)
        for k in range(8):                      # two full buckets + a partial one
            x.begin_step()
            ok &= loop.wire.data_ptr() == x.send[(k // K) & 1][k % K].data_ptr()
            loop.step_fused(rank, k)
            x.exchange()
            if k % K == K - 1:
                hdr, rows = x.latest()
                ok &= hdr.shape == (world * S, K)
                for g in range(world * S):
                    for j in range(K):
                        kk = k - (K - 1) + j
                        m = 1 + ((g + 3 * kk) % 7)
                        ok &= int(hdr["n_rows"][g, j]) == m and int(hdr["stream"][g, j]) == g and int(hdr["frame"][g, j]) == kk + 1
                        ok &= list(rows[g, j]["id"][:m]) == list(100000 * kk + 1000 * g + np.arange(m))
        x.flush()                               # steps 6, 7 of the third bucket
        hdr, rows = x.latest()
        for g in range(world * S):
            for j, kk in ((0, 6), (1, 7)):
                m = 1 + ((g + 3 * kk) % 7)
                ok &= int(hdr["n_rows"][g, j]) == m and int(hdr["frame"][g, j]) == kk + 1
        try:                                    # a step without begin_step() cannot be bucketed: refused, not silently stale
            loop.fill(rank, 99)
            x.exchange()
            ok = False
        except RuntimeError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucketed_per_frame_gather_world2_gloo():
    """TrackTableExchange(bucket=3): the tables of three consecutive time-steps in one all-gather, over two gloo ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_track_table_allgather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _FakeLoop:
    """What TrackTableExchange needs from a HotLoop, on CPU tensors."""

    def __init__(self, S, W, tcap, fused_step=False):
        import torch
        self.S, self.W, self.tcap, self.dev = S, W, tcap, torch.device("cpu")
        self.snap = torch.zeros(S, W, tcap, 64, dtype=torch.uint8)
        self.snap_n = torch.zeros(S, W, dtype=torch.int32)
        self.fused_step, self.wire, self._wire_ids = fused_step, None, (0, 0)      # HotLoop's one-launch step (window 1)

    def set_wire(self, wire, stream0=0, frame0=0):
        self.wire, self._wire_ids = wire, (stream0, frame0)

    def step_fused(self, rank, k):
        """What av_hot_step does with the wire buffer handed to it before the step: this step's tables in wire format,
        header.frame = frame0 + the stream's detector frame count (here: k + 1 for every stream)."""
        from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
        rows, n = self.fill(rank, k)
        msg = D.pack_wire(self.snap, self.snap_n, 0, 1, self._wire_ids[0], self._wire_ids[1] + k + 1)
        self.wire.copy_(msg.view(self.S, -1))
        return rows, n

    def fill(self, rank, k):
        """Deterministic tables for (rank, step k): returns the structured rows it wrote."""
        import torch
        rows = np.zeros((self.S, self.W, self.tcap), np.dtype(nat.TRACK_ROW_FIELDS))
        n = np.zeros((self.S, self.W), np.int32)
        for s in range(self.S):
            for f in range(self.W):
                g = rank * self.S + s
                m = 1 + ((g + f + 3 * k) % 7)
                n[s, f] = m
                r = rows[s, f]
                r["id"][:m] = 100000 * k + 1000 * g + 10 * f + np.arange(m)
                r["x1"][:m], r["y1"][:m] = 5 + g, 7 + f
                r["x2"][:m], r["y2"][:m] = 1200 + np.arange(m), 700 + k
                r["cls"][:m], r["age"][:m], r["hits"][:m], r["misses"][:m] = g % 8, 70000 + f, 3 + k, f % 5
                r["flags"][:m] = (np.arange(m) + k) % 2
                r["conf"][:m] = 0.75 + 0.001 * (g + f)
                r["vx"][:m], r["vy"][:m] = 0.5 * (g - 3), -1.5 * (f + 1)
                r["id"][m:] = -77                      # rows beyond the count are unspecified: must not travel
        self.snap.copy_(torch.as_tensor(rows.view(np.uint8).reshape(self.S, self.W, self.tcap, 64)))
        self.snap_n.copy_(torch.as_tensor(n))
        return rows, n


def _check_gather(hdr, rows, world, S, W, tcap, k, per_frame):
    ok = hdr.shape == (world * S, W if per_frame else 1)
    for g in range(world * S):
        for j, f in enumerate(range(W) if per_frame else [W - 1]):
            m = 1 + ((g + f + 3 * k) % 7)
            ok &= int(hdr["n_rows"][g, j]) == m and int(hdr["stream"][g, j]) == g and int(hdr["frame"][g, j]) == k * W + f
            r = rows[g, j]
            ok &= list(r["id"][:m]) == list(100000 * k + 1000 * g + 10 * f + np.arange(m)) and not r["id"][m:].any()
            ok &= int(r["x1"][0]) == 5 + g and int(r["y2"][0]) == 700 + k and int(r["age"][0]) == 70000 + f
            ok &= int(r["misses"][0]) == f % 5 and int(r["cls"][0]) == g % 8 and int(r["flags"][0]) == k % 2
            ok &= float(r["conf"][0]) == float(np.float32(0.75 + 0.001 * (g + f)))
            ok &= int(r["vx2"][0]) == g - 3 and int(r["vy2"][0]) == -3 * (f + 1)
    return bool(ok)


def _xchg_worker(rank, world, port, q):
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for per_frame in (False, True):
            S, W, tcap = 3, 4, 64
            loop = _FakeLoop(S, W, tcap)
            x = D.TrackTableExchange(loop, world, rank, per_frame=per_frame)
            ok &= x.bytes_per_step == S * (W if per_frame else 1) * (16 + 32 * tcap)
            held = []
            for k in range(4):                       # >= 3 windows: both buffers are re-used
                loop.fill(rank, k)
                buf = x.exchange()
                held.append((k, buf))
                hdr, rows = x.latest()
                ok &= _check_gather(hdr, rows, world, S, W, tcap, k, per_frame)
                if k >= 1:                           # the other buffer still holds the previous step's gather
                    h2, r2 = D.unpack_wire(held[k - 1][1], tcap)
                    ok &= _check_gather(h2, r2, world, S, W, tcap, k - 1, per_frame)
                ok &= buf.data_ptr() == x.recv[k & 1].data_ptr()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_track_table_exchange_class_world2_gloo():
    """TrackTableExchange itself (the class bench.py uses) over two gloo ranks: wire format, both gather modes,
    double-buffer order over four steps."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_xchg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_native_allgather_on_cpu_tensors_falls_back_with_a_warning():
    """TrackTableExchange(native=True) is an RCCL call on device buffers; on CPU tensors the class says so and gathers through
    torch.distributed (it used to clear the flag silently).  A loop that never called begin_step() is packed by exchange()."""
    import torch
    from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
    loop = _FakeLoop(2, 1, 64, fused_step=True)
    with pytest.warns(RuntimeWarning, match="needs device tensors"):
        x = D.TrackTableExchange(loop, 1, 0, per_frame=True, native=True)
    assert x.native is False and x.nccl is None and x.prepacked
    # pack_wire's two header.frame conventions: index within the run (no counters) / detector frame count (counters given)
    rows, n = loop.fill(0, 3)
    a = D.pack_wire(loop.snap, loop.snap_n, 0, 1, 10, 5)
    b = D.pack_wire(loop.snap, loop.snap_n, 0, 1, 10, 5, torch.tensor([40, 57], dtype=torch.int32))
    ha, _ = D.unpack_wire(a, 64)
    hb, rb = D.unpack_wire(b, 64)
    assert list(ha["frame"][:, 0]) == [5, 5] and list(hb["frame"][:, 0]) == [45, 62] and list(hb["stream"][:, 0]) == [10, 11]
    assert list(rb[1, 0]["id"][:n[1, 0]]) == list(rows[1, 0]["id"][:n[1, 0]])


def test_wire_layout_matches_header():
    from multimodal_autonomous_driving_perception_and_planning_amd import distributed as D
    assert np.dtype(D.WIRE_ROW_FIELDS).itemsize == 32 and np.dtype(D.WIRE_HDR_FIELDS).itemsize == 16
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "avhot.h")).read()
    assert "#define AV_WIRE_ROW_BYTES 32" in hdr and "#define AV_WIRE_HDR_BYTES 16" in hdr


def test_generator_vehicle_boxes_match_the_painted_frame():
    """Host metadata of the synthetic generator == what the (oracle == device) frame formula paints."""
    from multimodal_autonomous_driving_perception_and_planning_amd.generators import vehicle_boxes
    from oracle.lane_ref import synthetic_frame
    for h, w, stream, frame in ((720, 1280, 0, 0), (480, 640, 3, 7), (250, 333, 1, 26)):
        img = synthetic_frame(h, w, stream, frame)
        boxes = vehicle_boxes(h, w, stream, frame)
        assert 3 <= len(boxes) <= 6
        cover = np.full((h, w), -1)
        for i, (x1, y1, x2, y2, _) in enumerate(boxes):
            assert 0 <= x1 < x2 <= w and 0 <= y1 < y2 <= h
            cover[y1:y2, x1:x2] = i
        for i, (_, _, _, _, col) in enumerate(boxes):
            m = cover == i
            if m.any():
                assert (img[m] == np.array(col, np.uint8)).all(), (h, w, i)
        import data.generators.synthetic_data as shim          # the reference's import path resolves
        assert shim.vehicle_boxes is vehicle_boxes


def test_host_generators_match_the_oracle_formulas():
    """bench.py and the tools draw their synthetic inputs from the package (never from oracle/): same values."""
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion, synthetic_frame
    from oracle.harness_ref import ego_motion
    from oracle.lane_ref import synthetic_frame as oracle_frame
    assert np.array_equal(np.asarray(generate_ego_motion(64, seed=5)), ego_motion(64, seed=5))
    for args in ((120, 160, 1, 2), (250, 333, 1, 26)):
        assert np.array_equal(synthetic_frame(*args), oracle_frame(*args)), args


def test_only_tests_smoke_and_cpu_baseline_touch_the_oracle():
    """The oracle is test infrastructure: nothing in the package or the tools imports it; bench.py only in its
    cpu_baseline() leg and __graft_entry__ only in smoke()."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    pkg = os.path.join(root, "multimodal_autonomous_driving_perception_and_planning_amd")
    for base in (pkg, os.path.join(root, "tools"), os.path.join(root, "src"), os.path.join(root, "data")):
        for dp, _, files in os.walk(base):
            for f in files:
                if f.endswith(".py"):
                    assert not pat.search(open(os.path.join(dp, f)).read()), os.path.join(dp, f)
    bench = open(os.path.join(root, "bench.py")).read()
    hits = list(pat.finditer(bench))
    assert hits
    for m in hits:          # only inside the cpu_* baseline legs; the child of the nproc leg imports it by string
        assert bench[:m.start()].rsplit("\ndef ", 1)[1].startswith("cpu_"), bench[m.start():m.start() + 60]
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    for m in pat.finditer(entry):
        assert entry[:m.start()].rsplit("\ndef ", 1)[1].startswith("smoke(")


def test_step_flag_layout_matches_the_header_and_step_numbers_wrap():
    """The Python side allocates the overlapped steps' sequence flags: its size formula is the header's macro; step numbers are kept
    modulo 2^32 and cross the C ABI as signed ints."""
    import re
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "avhot.h")).read()
    m = re.search(r"#define AV_STEP_FLAG_INTS\(n_streams\) \((.*)\)\s*$", hdr, re.M)
    assert m, "AV_STEP_FLAG_INTS not found in include/avhot.h"
    for S in (1, 7, 64, 256):
        assert eval(m.group(1).replace("(n_streams)", str(S))) == nat.step_flag_ints(S)
    assert int(re.search(r"#define AV_STEP_MAX_DEPTH (\d+)", hdr).group(1)) == 4
    assert [nat.step_i32(v) for v in (0, 5, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 3)] == [0, 5, 2 ** 31 - 1, -2 ** 31, -1, 0, 3]
