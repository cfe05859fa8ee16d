"""The one-launch time-step (av_hot_step, BASELINE config 4 with window 1) against the four stage launches it replaces:
every output bit for bit, step after step, and against the CPU oracle of the loop (demo.py:97-120 call order)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    t = pytest.importorskip("torch")
    if not t.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return t


def _outputs(loop):
    loop.synchronize()
    r = loop.results()
    rows, n = loop.snapshots()
    hdr, trows, hist = loop.tracker_tables()
    out = dict(r)
    out.update(snap=rows.view(np.uint8), snap_n=n, hdr=hdr, trows=trows.view(np.uint8), hist=hist,
               kf=loop.kf_state.cpu().numpy(), plan_state=loop.plan_state.cpu().numpy(), fc=loop.frame_count.cpu().numpy())
    return out


def _same(a, b, where):
    assert a.keys() == b.keys()
    for k in a:
        x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
        assert x.shape == y.shape and x.dtype == y.dtype, (where, k)
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (where, k, int((x != y).sum()))


def test_fused_step_equals_the_four_stage_launches_bit_for_bit(torch):
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from oracle.harness_ref import run_stream
    S, steps = 9, 140                       # long enough for births, confirmations, deaths (max_age 30) and ring wrap-around (50)
    offs = [17 * s for s in range(S)]
    z = np.stack([run_stream(steps, frame_offset=offs[s], ego_seed=s)["z"] for s in range(S)])       # [S, steps, 4]
    fused = HotLoop(n_streams=S, window=1, fused_step=True)
    stage = HotLoop(n_streams=S, window=1, fused_step=False)
    assert fused.fused_step and not stage.fused_step
    wire = torch.zeros(S, int(nat.lib().av_wire_table_bytes(64)), dtype=torch.uint8, device=fused.dev)
    ref_wire = torch.zeros_like(wire)
    for lp in (fused, stage):
        lp.reset(frame_offsets=offs)
        # stream 3: a user-assigned, non-separable covariance -> the dense filter (flagged in kf_state[45] by the first step)
        kf = lp.kf_state.cpu().numpy()
        kf[3, 6 + 0 * 6 + 1] = kf[3, 6 + 1 * 6 + 0] = 0.25
        lp.kf_state.copy_(torch.as_tensor(kf))
    for t in range(steps):
        for lp in (fused, stage):
            lp.load_measurements(z[:, t:t + 1])
        fused.set_wire(wire, stream0=40, frame0=1000)
        fused.step(graph=False)
        stage.step(graph=False)
        if t % 7 == 0 or t > steps - 4:
            _same(_outputs(fused), _outputs(stage), "step %d" % t)
            nat.check(nat.lib().av_pack_tracks(stage.ctx.handle, stage._s, S, 1, 64, 0, 1, 40, 1000, nat.ptr(stage.snap),
                                               nat.ptr(stage.snap_n), nat.ptr(stage.frame_count), nat.ptr(ref_wire)))
            stage.synchronize()
            want = ref_wire.cpu().numpy().copy()
            # both stamp header.frame = frame0 + the stream's detector frame count (offset + steps so far)
            assert np.array_equal(want[:, 8:12], (1000 + np.asarray(offs, np.int32) + t + 1).astype("<i4").view(np.uint8).reshape(S, 4))
            assert np.array_equal(wire.cpu().numpy(), want), t
    out = _outputs(fused)
    assert out["kf"][3, 45] == 1.0 and out["kf"][0, 45] == 0.0          # the dense filter really ran for stream 3
    assert out["snap_n"].max() > 20 and out["hdr"][:, 1].min() > 40      # tables filled up, ids were issued
    # the same through a captured graph (one node), from a fresh state
    g = HotLoop(n_streams=S, window=1, fused_step=True)
    g.reset(frame_offsets=offs)
    s2 = HotLoop(n_streams=S, window=1, fused_step=False)
    s2.reset(frame_offsets=offs)
    for t in range(12):
        g.load_measurements(z[:, t:t + 1]), s2.load_measurements(z[:, t:t + 1])
        g.step(graph=True), s2.step(graph=True)
    _same(_outputs(g), _outputs(s2), "graph")


def test_fused_step_matches_the_cpu_oracle(torch):
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from oracle.harness_ref import run_stream
    S, steps = 3, 40
    offs = [0, 17, 34]
    want = [run_stream(steps, frame_offset=offs[s], ego_seed=s) for s in range(S)]
    loop = HotLoop(n_streams=S, window=1)
    assert loop.fused_step                                 # the default for window 1
    loop.reset(frame_offsets=offs)
    for t in range(steps):
        loop.load_measurements(np.stack([w["z"][t:t + 1] for w in want]))
        loop.step(sync=True)
        r = loop.results()
        rows, n = loop.snapshots()
        for s in range(S):
            assert np.array_equal(r["det_box"][s, 0], want[s]["det_box"][t]), (t, s)
            assert n[s, 0] == want[s]["n_live"][t] and np.array_equal(rows[s, 0]["id"][:n[s, 0]], want[s]["ids"][t][:n[s, 0]]), (t, s)
            np.testing.assert_allclose(r["vstate"][s, 0], want[s]["state"][t], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(r["cost"][s, 0], want[s]["cost"][t], rtol=1e-9)


def test_fused_step_refuses_shapes_it_is_not_built_for(torch):
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    with pytest.raises(ValueError):
        HotLoop(n_streams=2, window=4, fused_step=True)
    with pytest.raises(ValueError):
        HotLoop(n_streams=2, window=1, tcap=128, fused_step=True)
    assert not HotLoop(n_streams=2, window=1, tcap=128).fused_step        # falls back to the stage launches by itself


def _valid_rows(o):
    """Snapshot rows past a frame's live count are not written by the step (they keep whatever an earlier step left, and the two
    buffer sets of overlap=2 have different histories): compare the live rows only."""
    snap = o["snap"].copy()
    S, W = o["snap_n"].shape
    snap = snap.reshape(S, W, 64, -1)
    for s in range(S):
        for f in range(W):
            snap[s, f, o["snap_n"][s, f]:] = 0
    o["snap"] = snap
    return o


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_overlapped_steps_equal_the_serial_loop_bit_for_bit(torch, depth):
    """HotLoop(overlap=2): consecutive steps on two HIP streams, ordered per stream and role by the device-side sequence flags
    (av_hot_step_seq).  (a) enqueued back to back without any host synchronisation -- the case the flags exist for -- every
    step's wire table (one buffer per step) and the state after the last step equal the serial loop's; (b) synchronised step by
    step, every per-step output equals the serial loop's."""
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from oracle.harness_ref import run_stream
    S, steps = 64, 150
    offs = [17 * s for s in range(S)]
    z = np.stack([run_stream(steps, frame_offset=offs[s], ego_seed=s)["z"] for s in range(S)])       # [S, steps, 4]
    wb = int(nat.lib().av_wire_table_bytes(64))
    serial = HotLoop(n_streams=S, window=1)
    over = HotLoop(n_streams=S, window=1, overlap=depth)
    assert serial.fused_step and over.overlap == depth
    # (a) constant measurements (the step reads z from its buffer set; both sets are loaded once), no host synchronisation
    wires = {}
    for name, lp in (("serial", serial), ("over", over)):
        lp.reset(frame_offsets=offs)
        w = torch.zeros(steps, S, wb, dtype=torch.uint8, device=lp.dev)
        lp.load_measurements(z[:, :1], **({"all_sets": True} if lp.overlap > 1 else {}))
        for t in range(steps):
            lp.set_wire(w[t], stream0=40, frame0=1000)
            lp.enqueue_step()
        lp.synchronize()
        wires[name] = w.cpu().numpy()
    for t in range(steps):
        assert np.array_equal(wires["serial"][t], wires["over"][t]), ("wire table of step", t)
    _same(_valid_rows(_outputs(serial)), _valid_rows(_outputs(over)), "after %d unsynchronised steps" % steps)
    fl = over.seq_flags.cpu().numpy()
    assert fl[0:64 * S:32].tolist() == [steps] * (2 * S) and fl[64 * S] == 0
    # (b) fresh measurements every step, every output of every step
    for lp in (serial, over):
        lp.set_wire(None)
        lp.reset(frame_offsets=offs)
    for t in range(60):
        for lp in (serial, over):
            lp.load_measurements(z[:, t:t + 1])
            lp.enqueue_step()
        _same(_valid_rows(_outputs(serial)), _valid_rows(_outputs(over)), "step %d" % t)
    # (c) the launch loop in C (av_hot_steps_seq): fresh measurements for every step from one device tensor, every step's wire tables
    for lp in (serial, over):
        lp.reset(frame_offsets=offs)
    ws = torch.zeros(steps, S, wb, dtype=torch.uint8, device=serial.dev)
    for t in range(steps):
        serial.load_measurements(z[:, t:t + 1])
        serial.set_wire(ws[t], stream0=7, frame0=50)
        serial.enqueue_step()
    wo = torch.zeros_like(ws)
    zs = torch.as_tensor(np.ascontiguousarray(z.transpose(1, 0, 2))).to(over.dev)          # [steps, S, 4]
    over.set_wire(None, stream0=7, frame0=50)
    over.enqueue_steps(100, z_steps=zs[:100], wire_steps=wo[:100])
    over.enqueue_steps(steps - 100, z_steps=zs[100:], wire_steps=wo[100:])                  # an odd/even split: parity carries over
    _same(_valid_rows(_outputs(serial)), _valid_rows(_outputs(over)), "after the C launch loop")
    assert torch.equal(ws, wo)
    # the stage calls are not ordered across the two streams: refused, not raced
    with pytest.raises(RuntimeError, match="overlap=2"):
        over.enqueue_track()
    with pytest.raises(RuntimeError, match="overlap=2"):
        over.step(graph=True)
    with pytest.raises(ValueError, match="do not fit"):
        HotLoop(n_streams=128, window=1, overlap=3)


def test_overlapped_step_without_its_predecessor_faults_instead_of_hanging(torch, monkeypatch):
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    monkeypatch.setenv("AVHOT_STEP_SPIN", "2000")
    lp = HotLoop(n_streams=4, window=1, overlap=2)
    lp.reset()
    lp.enqueue_step()
    lp.synchronize()
    lp._seq = 5                              # steps 1..4 were never launched: step 5 waits for a counter that stays at 1
    lp.enqueue_step()
    with pytest.raises(RuntimeError, match="waited in vain"):
        lp.synchronize()
    assert lp.frame_count.cpu().numpy().tolist() == [1] * 4          # the faulted step did not run
    lp.reset()                               # clears the flags and the fault word
    lp.enqueue_step()
    lp.enqueue_step()
    lp.synchronize()
    assert lp.frame_count.cpu().numpy().tolist() == [2] * 4


@pytest.mark.parametrize("depth,start", [(4, 2 ** 31 - 40), (3, 2 ** 32 - 40), (2, 2 ** 32 - 40)])
def test_overlapped_step_numbers_wrap(torch, depth, start):
    """The library's step numbers are 32-bit (three hours of stepping at 5 us per step): a loop whose counter is put just below
    2^31 / 2^32 steps on through the wrap with the serial loop's results (at depth 3 the stream of a step changes
    discontinuously at 2^32: two consecutive steps on one stream, still in order)."""
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from oracle.harness_ref import run_stream
    S, steps = 16, 100
    offs = [17 * s for s in range(S)]
    z = np.stack([run_stream(2, frame_offset=offs[s], ego_seed=s)["z"][:1] for s in range(S)])
    serial = HotLoop(n_streams=S, window=1)
    over = HotLoop(n_streams=S, window=1, overlap=depth)
    for lp in (serial, over):
        lp.reset(frame_offsets=offs)
    over._seq = start & 0xFFFFFFFF
    fl = over.seq_flags.cpu().numpy().astype(np.int64)
    fl[0:64 * S:32] = start                                                    # the counters: `start` steps done
    fl[64 * S + 32:65 * S + 32] = np.asarray(offs, np.int64) - start           # frame count at "reset" = now - start
    over.seq_flags.copy_(torch.as_tensor((fl & 0xFFFFFFFF).astype(np.uint32).view(np.int32)))
    torch.cuda.synchronize()
    serial.load_measurements(z)
    over.load_measurements(z, all_sets=True)
    over.enqueue_steps(60)
    for _ in range(steps - 60):
        over.enqueue_step()
    for _ in range(steps):
        serial.enqueue_step()
    assert over._seq == (start + steps) & 0xFFFFFFFF
    _same(_valid_rows(_outputs(serial)), _valid_rows(_outputs(over)), "through the wrap at %d" % start)


def test_hot_loop_tune_streams_changes_the_streams_not_the_results(torch):
    """HotLoop.tune_streams() (what bench.py calls first): measures a dozen stream sets, keeps one, resets the loop -- the steps after
    it equal the serial loop's."""
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from oracle.harness_ref import run_stream
    S, steps = 8, 40
    offs = [17 * s for s in range(S)]
    z = np.stack([run_stream(2, frame_offset=offs[s], ego_seed=s)["z"][:1] for s in range(S)])
    serial = HotLoop(n_streams=S, window=1)
    over = HotLoop(n_streams=S, window=1, overlap=4)
    tried = over.tune_streams(pool=8, candidates=6, steps=100)
    assert len(tried) == 6 and all(t > 0 for t in tried)
    assert len({st.cuda_stream for st in over._pstreams}) == 4 and over.stream is over._pstreams[0]
    assert HotLoop(n_streams=S, window=1).tune_streams() == []                  # nothing to choose for serial launches
    for lp in (serial, over):
        lp.reset(frame_offsets=offs)
    serial.load_measurements(z)
    over.load_measurements(z, all_sets=True)
    over.enqueue_steps(steps)
    for _ in range(steps):
        serial.enqueue_step()
    # (reset() rewinds the tables' headers: the rows and history rings past the live count keep what the measured runs left there,
    # so the persistent tables are compared through the snapshot rows of the live tracks)
    a, b = _valid_rows(_outputs(serial)), _valid_rows(_outputs(over))
    for k in ("trows", "hist"):
        a.pop(k), b.pop(k)
    _same(a, b, "after tune_streams")
