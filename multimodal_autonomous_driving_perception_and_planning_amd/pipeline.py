"""Batched hot loop: S independent video streams x W frames per launch, all state resident in HBM.

This is the host-side driver of the simulated-detection configuration (BASELINE configs 2/4/5):

    detect (av_simdet_generate) -> track (av_tracker_update)          side stream
                                   Kalman (av_kf_step) -> plan (av_planner_plan)   main stream

in the reference's per-frame call order (demo.py:97-120).  Tracker and Kalman/planner are
independent (the planner never consumes tracks), so they run as two branches
of one fork/join, optionally captured into a hipGraph.  PyTorch only provides device memory and
the stream; every kernel is in libavhot.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat


class HotLoop:
    def __init__(self, n_streams=1, window=1, h=720, w=1280, tcap=64, dcap=8, device=0,
                 tracker_kw=None, kf_kw=None, planner_kw=None, keep_waypoints=True, keep_snapshots=True,
                 ctx=None, fused_step=None, overlap=1):
        """fused_step: with window 1, run a time-step as ONE launch (av_hot_step: role-split workgroups running the stage
        kernels' own device code, same results bit for bit) instead of the four stage launches.  None = whenever the
        configuration allows it (window 1, tcap 64, dcap 7..8, iou_threshold > 0).
        overlap=2 (fused step only): consecutive steps are launched alternately on two HIP streams and ordered per stream and
        role on the device (av_hot_step_seq), so step t + 1 starts while step t's planner is still writing.  The per-step
        buffers (det_*, snap, snap_n, det2trk, z, vstate, plan_state, wp, cost, order) then exist twice; the attributes always
        name the set of the step enqueued LAST (for `z`: the set the NEXT step will read), and that set is next written by
        the step after the next one.  Same results as overlap=1 bit for bit."""
        if not torch.cuda.is_available():
            raise RuntimeError("HotLoop needs a HIP device; this package has no CPU path")
        self.S, self.W, self.h, self.w, self.tcap, self.dcap = n_streams, window, h, w, tcap, dcap
        self.dev = torch.device("cuda", device)
        self.ctx = ctx or nat.Context(device)
        self.L = nat.lib()
        tk = dict(iou_threshold=0.3, max_age=30, min_hits=3, trajectory_length=50)
        tk.update(tracker_kw or {})
        self.tcfg = nat.TrackerCfg(**tk)
        kk = dict(dt=0.033, process_noise=0.1, measurement_noise=1.0)
        kk.update(kf_kw or {})
        self.kcfg = nat.KfCfg(**kk)
        pk = dict(planning_horizon=5.0, dt=0.1, num_samples=7, w_lateral=1.0, w_velocity=0.5,
                  w_acceleration=0.3, w_curvature=0.4)
        pk.update(planner_kw or {})
        self.pcfg = nat.PlannerCfg(reserved=0, **pk)
        nat.check(self.L.av_planner_configure(self.ctx.handle, C.byref(self.pcfg)))
        n, c = C.c_int(), C.c_int()
        nat.check(self.L.av_planner_dims(self.ctx.handle, C.byref(n), C.byref(c)))
        self.n_points, self.n_cand = n.value, c.value

        S, W, d = n_streams, window, self.dev
        i32, f64 = torch.int32, torch.float64
        self.frame_count = torch.zeros(S, dtype=i32, device=d)
        self.det_n = torch.zeros(S, W, dtype=i32, device=d)
        self.det_box = torch.zeros(S, W, dcap, 4, dtype=i32, device=d)
        self.det_cls = torch.zeros(S, W, dcap, dtype=i32, device=d)
        self.det_conf = torch.zeros(S, W, dcap, dtype=f64, device=d)
        self.det_status = torch.zeros(S, dtype=i32, device=d)
        self.trk_bytes = int(self.L.av_tracker_state_bytes(tcap, self.tcfg.trajectory_length))
        self.trk_state = torch.zeros(S, self.trk_bytes, dtype=torch.uint8, device=d)
        self.keep_snapshots = keep_snapshots
        self.snap = torch.zeros(S, W, tcap, nat.TRACK_ROW_BYTES, dtype=torch.uint8, device=d) if keep_snapshots else None
        self.snap_n = torch.zeros(S, W, dtype=i32, device=d) if keep_snapshots else None
        self.det2trk = torch.zeros(S, W, dcap, dtype=i32, device=d)
        self.z = torch.zeros(S, W, 4, dtype=f64, device=d)
        self.kf_state = torch.zeros(S, nat.KF_STATE_DOUBLES, dtype=f64, device=d)
        self.vstate = torch.zeros(S, W, nat.VSTATE_DOUBLES, dtype=f64, device=d)
        self.plan_state = torch.zeros(S, W, 4, dtype=f64, device=d)
        self.keep_waypoints = keep_waypoints
        self.wp = (torch.zeros(S * W, self.n_cand, self.n_points, nat.WP_DOUBLES, dtype=f64, device=d)
                   if keep_waypoints else None)
        self.cost = torch.zeros(S * W, self.n_cand, dtype=f64, device=d)
        self.order = torch.zeros(S * W, self.n_cand, dtype=i32, device=d)
        self.stream = torch.cuda.Stream(device=d)
        self.graph_id = None
        self._graphs = {}
        can_fuse = window == 1 and tcap == 64 and 7 <= dcap <= 8 and self.tcfg.iou_threshold > 0
        if fused_step and not can_fuse:
            raise ValueError("fused_step needs window 1, tcap 64, dcap 7..8 and iou_threshold > 0")
        self.fused_step = can_fuse if fused_step is None else bool(fused_step)
        self.wire = None                  # set_wire(): the fused step also writes every stream's table in wire format
        self._wire_ids = (0, 0)
        self.overlap = int(overlap)
        if self.overlap not in (1, 2, 3, 4):
            raise ValueError("overlap is 1 .. 4")
        if self.overlap > 1:
            if not self.fused_step:
                raise ValueError("overlap=2 needs the fused step (window 1, tcap 64, dcap 7..8, iou_threshold > 0)")
            names = [k for k in self._PER_STEP if getattr(self, k) is not None]
            self._sets = [{k: getattr(self, k) for k in names}] + [{k: torch.zeros_like(getattr(self, k)) for k in names}
                                                                   for _ in range(self.overlap - 1)]
            self._pstreams = [self.stream] + [torch.cuda.Stream(device=d) for _ in range(self.overlap - 1)]
            self.seq_flags = torch.zeros(nat.step_flag_ints(S), dtype=i32, device=d)
            self._seq, self._stepped = 0, False
            self._csets = None
            self.reset()
            try:                              # (one step now: the library refuses a depth whose launches would not all be resident)
                self._enqueue_step_seq()
                self.synchronize()
            except RuntimeError as e:
                raise ValueError("overlap=%d: %s" % (self.overlap, e)) from None
        self.reset()

    _PER_STEP = ("det_n", "det_box", "det_cls", "det_conf", "snap", "snap_n", "det2trk", "z", "vstate", "plan_state", "wp",
                 "cost", "order")

    # ------------------------------------------------------------------------------------------
    @property
    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    _i32 = staticmethod(nat.step_i32)

    def _serial_only(self, what):
        if self.overlap != 1:
            raise RuntimeError("%s is not available with overlap=2.. (only the one-launch step is ordered across the loop's streams)" % what)

    def reset(self, frame_offsets=None):
        """Resets every stream (tracker.reset(), state_estimator.reset(), detector.reset())."""
        h, L = self.ctx.handle, self.L
        if self.overlap > 1:
            for st in self._pstreams:
                st.synchronize()
        nat.check(L.av_tracker_reset(h, self._s, self.S, self.tcap, self.tcfg.trajectory_length, nat.ptr(self.trk_state)))
        nat.check(L.av_kf_reset(h, self._s, self.S, nat.ptr(self.kf_state)))
        with torch.cuda.stream(self.stream):
            if frame_offsets is None:
                self.frame_count.zero_()
            else:
                self.frame_count.copy_(torch.as_tensor(np.asarray(frame_offsets, np.int32)), non_blocking=False)
            if self.overlap > 1:
                self.seq_flags.zero_()
                self.seq_flags[64 * self.S + 32:65 * self.S + 32].copy_(self.frame_count)
        if self.overlap > 1:
            self._seq, self._stepped = 0, False
            self.__dict__.update(self._sets[0])
        self.stream.synchronize()

    def load_measurements(self, z, all_sets=False):
        """z: float64 [S, W, 4] ego measurements for the next window (host array).  overlap=2: for the next STEP (its buffer
        set); all_sets=True: for every step from now on (both sets)."""
        if self.overlap > 1:                  # the next step's set, on the next step's stream
            zt = torch.as_tensor(np.ascontiguousarray(z, np.float64)).view(self.S, self.W, 4)
            for k in (range(self.overlap) if all_sets else (self._seq % self.overlap,)):
                with torch.cuda.stream(self._pstreams[k]):
                    self._sets[k]["z"].copy_(zt)
                self._pstreams[k].synchronize()
            self.z = self._sets[self._seq % self.overlap]["z"]
            return
        with torch.cuda.stream(self.stream):
            self.z.copy_(torch.as_tensor(np.ascontiguousarray(z, np.float64)).view(self.S, self.W, 4))
        self.stream.synchronize()

    # ---- individual stages (enqueue only) --------------------------------------------------------
    def enqueue_detect(self, stream=None):
        self._serial_only("enqueue_detect")
        nat.check(self.L.av_simdet_generate(self.ctx.handle, stream or self._s, self.S, self.W, self.h, self.w,
                                            self.dcap, nat.ptr(self.frame_count), nat.ptr(self.det_n),
                                            nat.ptr(self.det_box), nat.ptr(self.det_cls), nat.ptr(self.det_conf),
                                            nat.ptr(self.det_status)))

    def enqueue_track(self, stream=None):
        self._serial_only("enqueue_track")
        nat.check(self.L.av_tracker_update(self.ctx.handle, stream or self._s, C.byref(self.tcfg), self.S, self.W,
                                           self.dcap, nat.ptr(self.det_n), nat.ptr(self.det_box),
                                           nat.ptr(self.det_cls), nat.ptr(self.det_conf), self.tcap,
                                           nat.ptr(self.trk_state), nat.ptr(self.snap), nat.ptr(self.snap_n),
                                           nat.ptr(self.det2trk)))

    def enqueue_kf(self, stream=None):
        self._serial_only("enqueue_kf")
        nat.check(self.L.av_kf_step(self.ctx.handle, stream or self._s, C.byref(self.kcfg), self.S, self.W,
                                    nat.ptr(self.z), None, nat.ptr(self.kf_state), nat.ptr(self.vstate),
                                    nat.ptr(self.plan_state)))

    def enqueue_maneuver(self, stream=None, lane_offset=None):
        """Maneuver tags of every frame of the window from the Kalman output (ManeuverDetector.detect,
        maneuver_detector.py:105-262); call after enqueue_kf on the same stream.  Results: self.maneuver
        (uint8 view of av_maneuver_row [S][W])."""
        self._serial_only("enqueue_maneuver")
        if not hasattr(self, "mv_state"):
            self.mv_state = torch.zeros(self.S, nat.MANEUVER_STATE_DOUBLES, dtype=torch.float64, device=self.dev)
            self.maneuver = torch.zeros(self.S, self.W, nat.MANEUVER_ROW_BYTES, dtype=torch.uint8, device=self.dev)
        nat.check(self.L.av_maneuver_detect(self.ctx.handle, stream or self._s, self.S, self.W, nat.ptr(self.vstate),
                                            nat.ptr(lane_offset), nat.ptr(self.mv_state), nat.ptr(self.maneuver)))

    def enqueue_interactions(self, stream=None, frame_shape=None, class_names=None):
        """Interaction tags of every frame of the window from the tracker's snapshot tables and the Kalman output
        (InteractionDetector.detect, interaction_detector.py:132-222); call where both are complete (after the
        join).  Results: self.inter_rows (av_interaction_row [S][W][tcap]), self.inter_summary ([S][W])."""
        self._serial_only("enqueue_interactions")
        from .perception.detector import ObjectDetector
        from .tagging.interaction_detector import interaction_cfg
        if self.tcap != 64 or self.snap is None:
            raise RuntimeError("enqueue_interactions needs keep_snapshots=True and tcap 64")
        if not hasattr(self, "inter_state"):
            self.inter_state = torch.zeros(self.S * int(self.L.av_interaction_state_bytes(self.tcap)), dtype=torch.uint8,
                                           device=self.dev)
            nat.check(self.L.av_interaction_reset(self.ctx.handle, stream or self._s, self.S, self.tcap, nat.ptr(self.inter_state)))
            self.inter_rows = torch.zeros(self.S, self.W, self.tcap, nat.INTERACTION_ROW_BYTES, dtype=torch.uint8, device=self.dev)
            self.inter_summary = torch.zeros(self.S, self.W, nat.INTERACTION_SUMMARY_BYTES, dtype=torch.uint8, device=self.dev)
        cfg = interaction_cfg(frame_shape or (self.h, self.w), class_names or [ObjectDetector.CLASSES[k] for k in range(8)])
        nat.check(self.L.av_interaction_detect(self.ctx.handle, stream or self._s, C.byref(cfg), self.S, self.W, self.tcap,
                                               nat.ptr(self.snap), nat.ptr(self.snap_n), nat.ptr(self.vstate), None, None,
                                               nat.ptr(self.inter_state), nat.ptr(self.inter_rows), nat.ptr(self.inter_summary)))

    def enqueue_bev(self, stream=None, frame=None, n_candidates=10):
        """BEV panels of every stream (BEVRenderer.render, bev_renderer.py:286-348) for one frame of the window, built and
        painted on the device from the tables the step left in HBM; call where tracker, Kalman and planner outputs are
        complete (after the join).  Results: self.bev (uint8 [S, 600, 600, 3]).  The trails come from the tracker's
        history rings, i.e. they are those of the window's last frame (the default)."""
        self._serial_only("enqueue_bev")
        from .visualization.bev_renderer import BEVRenderer
        if not (self.keep_waypoints and self.keep_snapshots):
            raise RuntimeError("enqueue_bev needs keep_waypoints=True and keep_snapshots=True")
        if not hasattr(self, "bev"):
            r = BEVRenderer(device=self.dev.index)
            self._bev_cfg = nat.BevCfg(r.width, r.height, r.pixels_per_meter, r.x_range[0], r.x_range[1], r.y_range[0], r.y_range[1],
                                       n_candidates, 0)
            self._bev_cap = int(self.L.av_bev_prim_cap(C.byref(self._bev_cfg), self.tcap, self.n_points))
            self._bev_base = torch.as_tensor(r.create_base_image()).to(self.dev)
            self.bev = torch.empty(self.S, r.height, r.width, 3, dtype=torch.uint8, device=self.dev)
            self._bev_prims = torch.zeros(self.S, self._bev_cap, nat.PRIM_BYTES, dtype=torch.uint8, device=self.dev)
            self._bev_n = torch.zeros(self.S, dtype=torch.int32, device=self.dev)
        st = stream or self._s
        f = self.W - 1 if frame is None else frame
        # the base image is copied on the stream the kernels run on (a caller-supplied stream must not race with it)
        ts = self.stream if stream is None else torch.cuda.ExternalStream(stream.value if hasattr(stream, "value") else int(stream), device=self.dev)
        with torch.cuda.stream(ts):
            self.bev.copy_(self._bev_base.unsqueeze(0).expand_as(self.bev))
        nat.check(self.L.av_bev_build(self.ctx.handle, st, C.byref(self._bev_cfg), self.S, self.W, f, self.tcap,
                                      self.tcfg.trajectory_length, nat.ptr(self.snap), nat.ptr(self.snap_n), nat.ptr(self.trk_state),
                                      nat.ptr(self.vstate), nat.ptr(self.wp), nat.ptr(self.order), nat.ptr(self._bev_prims),
                                      self._bev_cap, nat.ptr(self._bev_n)))
        nat.check(self.L.av_raster_draw(self.ctx.handle, st, self.S, self._bev_cfg.height, self._bev_cfg.width, nat.ptr(self.bev),
                                        nat.ptr(self._bev_prims), self._bev_cap, nat.ptr(self._bev_n), None, 0))

    def enqueue_plan(self, stream=None):
        self._serial_only("enqueue_plan")
        nat.check(self.L.av_planner_plan(self.ctx.handle, stream or self._s, self.S * self.W,
                                         nat.ptr(self.plan_state), None, 0, None, 0, nat.ptr(self.wp),
                                         nat.ptr(self.cost), nat.ptr(self.order)))

    def set_wire(self, wire, stream0=0, frame0=0):
        """Fused step only: `wire` (uint8 device tensor [S, av_wire_table_bytes(tcap)], or None) receives every stream's
        track table in the all-gather's wire format from the same launch; header.stream = stream0 + s, header.frame =
        frame0 + the stream's detector frame count after the step."""
        if wire is not None and not (self.fused_step and self.keep_snapshots):
            raise RuntimeError("set_wire needs the fused step and keep_snapshots=True")
        self.wire, self._wire_ids = wire, (int(stream0), int(frame0))

    def enqueue_step_fused(self, stream=None):
        """Window 1: detect + track + Kalman + plan of one frame of every stream as ONE launch."""
        if self.overlap > 1:
            return self._enqueue_step_seq(stream)
        nat.check(self.L.av_hot_step(self.ctx.handle, stream or self._s, C.byref(self.tcfg), C.byref(self.kcfg), self.S, self.h,
                                     self.w, self.dcap, self.tcap, nat.ptr(self.frame_count), nat.ptr(self.det_n),
                                     nat.ptr(self.det_box), nat.ptr(self.det_cls), nat.ptr(self.det_conf),
                                     nat.ptr(self.det_status), nat.ptr(self.trk_state), nat.ptr(self.snap), nat.ptr(self.snap_n),
                                     nat.ptr(self.det2trk), nat.ptr(self.z), nat.ptr(self.kf_state), nat.ptr(self.vstate),
                                     nat.ptr(self.plan_state), nat.ptr(self.wp), nat.ptr(self.cost), nat.ptr(self.order),
                                     nat.ptr(self.wire), self._wire_ids[0], self._wire_ids[1]))

    def _enqueue_step_seq(self, stream=None):
        """overlap=D: step number self._seq on stream seq % D with buffer set seq % D, ordered behind step seq - 1 per stream and
        role by the sequence flags (av_hot_step_seq)."""
        if stream is not None:
            raise RuntimeError("overlap=2.. launches on the loop's own streams")
        k = self._seq % self.overlap         # (_seq is kept modulo 2^32: the library's step numbers are 32-bit and wrap)
        b = self._sets[k]
        self.__dict__.update(b)              # the attributes name the set of the step enqueued last
        nat.check(self.L.av_hot_step_seq(self.ctx.handle, C.c_void_p(self._pstreams[k].cuda_stream), C.byref(self.tcfg), C.byref(self.kcfg),
                                         self.S, self.h, self.w, self.dcap, self.tcap, nat.ptr(self.frame_count), nat.ptr(b["det_n"]),
                                         nat.ptr(b["det_box"]), nat.ptr(b["det_cls"]), nat.ptr(b["det_conf"]),
                                         nat.ptr(self.det_status), nat.ptr(self.trk_state), nat.ptr(b.get("snap")), nat.ptr(b.get("snap_n")),
                                         nat.ptr(b["det2trk"]), nat.ptr(b["z"]), nat.ptr(self.kf_state), nat.ptr(b["vstate"]),
                                         nat.ptr(b["plan_state"]), nat.ptr(b.get("wp")), nat.ptr(b["cost"]), nat.ptr(b["order"]),
                                         nat.ptr(self.wire), self._wire_ids[0], self._wire_ids[1], nat.ptr(self.seq_flags), self._i32(self._seq), self.overlap))
        self._seq = (self._seq + 1) & 0xFFFFFFFF
        self._stepped = True

    def enqueue_steps(self, n_steps, z_steps=None, wire_steps=None):
        """overlap=2: n_steps consecutive steps enqueued by one library call (av_hot_steps_seq: the launch loop in C).
        z_steps: None (every step reads its buffer set's z) or a float64 device tensor [n_steps, S, 4], the measurements of each
        step; wire_steps: None or a uint8 device tensor [n_steps, S, av_wire_table_bytes(tcap)] that receives every step's
        wire tables (set_wire's stream0 / frame0 apply)."""
        if self.overlap < 2:
            raise RuntimeError("enqueue_steps needs overlap=2..")
        if n_steps <= 0:
            return
        if self._csets is None:
            def cset(b):
                m = {"wp": "waypoints"}
                return nat.StepSet(**{m.get(k, k): (b[k].data_ptr() if b.get(k) is not None else None) for k in self._PER_STEP})
            self._csets = (nat.StepSet * self.overlap)(*[cset(b) for b in self._sets])
            self._cstreams = (C.c_void_p * self.overlap)(*[st.cuda_stream for st in self._pstreams])
        if z_steps is not None and (z_steps.dtype != torch.float64 or z_steps.numel() != n_steps * self.S * 4 or not z_steps.is_contiguous()):
            raise ValueError("z_steps: contiguous float64 [n_steps, S, 4]")
        if wire_steps is not None:
            wb = int(self.L.av_wire_table_bytes(self.tcap))
            if not (self.keep_snapshots and wire_steps.dtype == torch.uint8 and wire_steps.numel() == n_steps * self.S * wb and wire_steps.is_contiguous()):
                raise ValueError("wire_steps: contiguous uint8 [n_steps, S, %d] (and keep_snapshots=True)" % wb)
        nat.check(self.L.av_hot_steps_seq(self.ctx.handle, self.overlap, self._cstreams,
                                          C.byref(self.tcfg), C.byref(self.kcfg), self.S, self.h, self.w, self.dcap, self.tcap,
                                          nat.ptr(self.frame_count), nat.ptr(self.det_status), nat.ptr(self.trk_state), nat.ptr(self.kf_state),
                                          self._csets, nat.ptr(z_steps), nat.ptr(wire_steps),
                                          self._wire_ids[0], self._wire_ids[1], nat.ptr(self.seq_flags), self._i32(self._seq), int(n_steps)))
        self._seq = (self._seq + int(n_steps)) & 0xFFFFFFFF
        self._stepped = True
        self.__dict__.update(self._sets[((self._seq - 1) & 0xFFFFFFFF) % self.overlap])

    def tune_streams(self, pool=8, candidates=12, steps=600):
        """overlap=D: pick the D HIP streams the overlapped steps run on.  The runtime serves a process's streams from a few hardware
        queues, assigned by the process's whole stream history; launches on streams that share a queue do not overlap, and the same
        64-stream loop at depth 4 was measured at 4.9 us per step in one process and 10 us in another.  There is no way to ask for a
        queue, so the loop measures: `steps` steps (one library call) on each of `candidates` sets of D streams drawn from `pool`
        fresh ones, and keeps the fastest set.  Call it before reset() / loading state: the measured steps advance the streams'
        state, and the loop is reset afterwards.  -> us per step of every candidate set."""
        import time
        if self.overlap < 2:
            return []
        self.synchronize(check=False)
        D = self.overlap
        fresh = [torch.cuda.Stream(device=self.dev) for _ in range(max(pool, D))]
        rs = np.random.RandomState(12345)
        sets = [list(self._pstreams)] + [fresh[i:i + D] for i in range(0, len(fresh) - D + 1, D)]
        while len(sets) < candidates:
            sets.append([fresh[i] for i in sorted(rs.choice(len(fresh), D, replace=False))])
        tried = []
        for cand in sets:
            self._pstreams, self.stream = list(cand), cand[0]
            self._csets = None
            self.reset()
            self.enqueue_steps(64)
            self.synchronize()
            t0 = time.perf_counter()
            self.enqueue_steps(steps)
            self.synchronize()
            tried.append(((time.perf_counter() - t0) / steps * 1e6, cand))
        best = min(tried, key=lambda x: x[0])
        self._pstreams, self.stream = list(best[1]), best[1][0]
        self._csets = None
        self.reset()
        return [round(t, 2) for t, _ in tried]

    def step_stream(self):
        """The torch stream the step enqueued last runs on (overlap=2 alternates between two)."""
        return self._pstreams[((self._seq - 1) & 0xFFFFFFFF) % self.overlap] if self.overlap > 1 and self._stepped else self.stream

    def enqueue_step(self):
        """One window of the whole loop: fork{detect; track} || {kf; plan}; join.  Detections only feed the
        tracker, so both sit on the side stream and the Kalman/planner chain starts at once.  With window 1 and
        fused_step the whole step is one launch instead."""
        if self.fused_step:
            return self.enqueue_step_fused()
        h, L, s = self.ctx.handle, self.L, self._s
        nat.check(L.av_fork(h, s))
        self.enqueue_detect(self.ctx.side_stream)
        self.enqueue_track(self.ctx.side_stream)
        self.enqueue_kf()
        self.enqueue_plan()
        nat.check(L.av_join(h, s))

    def capture(self):
        """Capture enqueue_step() into a hipGraph (replayed by step(graph=True)).  (overlap=2: not available -- a graph would
        replay one step number.)  The fused step bakes the wire buffer's
        address into its kernel arguments, so graphs are kept per wire buffer (the exchange alternates between two)."""
        self._serial_only("capture()")
        gid = C.c_int(-1)
        nat.check(self.L.av_graph_begin(self.ctx.handle, self._s))
        try:
            self.enqueue_step()
        finally:
            nat.check(self.L.av_graph_end(self.ctx.handle, self._s, C.byref(gid)))
        self.graph_id = gid.value
        self._graphs[self._graph_key()] = gid.value
        return self.graph_id

    def _graph_key(self):
        return (self.wire.data_ptr(), self._wire_ids) if self.wire is not None else None

    def step(self, graph=False, sync=False):
        if graph:
            gid = self._graphs.get(self._graph_key())
            if gid is None:
                gid = self.capture()
            nat.check(self.L.av_graph_launch(self.ctx.handle, gid, self._s))
        else:
            self.enqueue_step()
        if sync:
            self.synchronize()

    def synchronize(self, check=True):
        if self.overlap > 1:
            for st in self._pstreams:
                st.synchronize()
            if check and int(self.seq_flags[64 * self.S].item()) != 0:
                raise RuntimeError("HotLoop(overlap=%d): a step waited in vain for its predecessor (sequence flags: fault word set); "
                                   "the state is no longer that of a serial run -- reset()" % self.overlap)
            return
        self.stream.synchronize()

    # ---- host views of the last window -------------------------------------------------------------
    def snapshots(self):
        """-> (rows structured array [S,W,tcap], n [S,W]) for the last window."""
        self.synchronize()
        raw = self.snap.cpu().numpy()
        rows = raw.view(np.dtype(nat.TRACK_ROW_FIELDS)).reshape(self.S, self.W, self.tcap)
        return rows, self.snap_n.cpu().numpy()

    def tracker_tables(self):
        """Persistent per-stream state: (hdr int32[S,16], rows [S,tcap], hist float64[S,tcap,L,4])."""
        self.synchronize()
        raw = self.trk_state.cpu().numpy()
        L = self.tcfg.trajectory_length
        hdr = raw[:, :nat.TRACKER_HDR_BYTES].copy().view(np.int32)
        ro = nat.TRACKER_HDR_BYTES
        rows = raw[:, ro:ro + self.tcap * 64].copy().view(np.dtype(nat.TRACK_ROW_FIELDS)).reshape(self.S, self.tcap)
        hist = raw[:, ro + self.tcap * 64:].copy().view(np.float64).reshape(self.S, self.tcap, L, 4)
        return hdr, rows, hist

    def results(self):
        self.synchronize()
        out = dict(det_n=self.det_n.cpu().numpy(), det_box=self.det_box.cpu().numpy(),
                   det_cls=self.det_cls.cpu().numpy(), det_conf=self.det_conf.cpu().numpy(),
                   det2trk=self.det2trk.cpu().numpy(), vstate=self.vstate.cpu().numpy(),
                   cost=self.cost.cpu().numpy().reshape(self.S, self.W, self.n_cand),
                   order=self.order.cpu().numpy().reshape(self.S, self.W, self.n_cand))
        if self.keep_waypoints:
            out["wp"] = self.wp.cpu().numpy().reshape(self.S, self.W, self.n_cand, self.n_points, 6)
        return out

    # algorithmic HBM bytes of one planner launch: per start state
    # 32 B read + C*n*48 B waypoints + C*8 B costs + C*4 B order
    def planner_bytes_per_state(self):
        c, n = self.n_cand, self.n_points
        return 32 + c * n * 48 + c * 8 + c * 4


class PerceptionLoop:
    """BASELINE config 3: S camera streams, frames generated on the device, YOLO-mode detector (MFMA conv
    path) + lane detector per frame.  Everything stays in HBM; one enqueue per stage per step."""

    def __init__(self, n_streams=16, h=720, w=1280, device=0, model="random:0", max_segments=512, ctx=None, precision="fp16"):
        if not torch.cuda.is_available():
            raise RuntimeError("PerceptionLoop needs a HIP device; this package has no CPU path")
        from .perception.yolo import MAX_DET, YoloV8n, conv_specs
        self.S, self.h, self.w, self.ms = n_streams, h, w, max_segments
        self.dev = torch.device("cuda", device)
        self.L = nat.lib()
        self.yolo = YoloV8n(model, device=device, batch=n_streams, precision=precision)
        self.ctx = self.yolo._dev.ctx
        self.yolo._prepare(h, w)
        # the detector chain is the critical path when the lane chain runs beside it: higher queue priority
        self.stream = torch.cuda.Stream(device=self.dev, priority=-1)
        S, d = n_streams, self.dev
        self.frames = torch.empty(S, h, w, 3, dtype=torch.uint8, device=d)
        self.ws = torch.empty(int(self.L.av_lane_workspace_bytes(S, h, w, max_segments)), dtype=torch.uint8, device=d)
        nat.check(self.L.av_lane_workspace_init(self.ctx.handle, self._s, S, h, w, max_segments, nat.ptr(self.ws)))
        self.lane_state = torch.zeros(S, 8, dtype=torch.float64, device=d)
        self.poly = torch.zeros(S, 2, 3, dtype=torch.float64, device=d)
        self.pts = torch.zeros(S, 2, 50, 2, dtype=torch.int32, device=d)
        self.info = torch.zeros(S, 8, dtype=torch.int32, device=d)
        self.conf = torch.zeros(S, 2, dtype=torch.float64, device=d)
        self.det_n = torch.zeros(S, dtype=torch.int32, device=d)
        self.det_box = torch.zeros(S, MAX_DET, 4, dtype=torch.float32, device=d)
        self.det_conf = torch.zeros(S, MAX_DET, dtype=torch.float32, device=d)
        self.det_cls = torch.zeros(S, MAX_DET, dtype=torch.int32, device=d)
        self.lcfg = nat.LaneCfg(50, 50, 150, max_segments, 0.7)
        self.max_det = MAX_DET
        net_h, net_w, _ = self.yolo.dims()
        # 2*MACs of every convolution at the letterboxed resolution (the figure the MFMA roofline is priced on)
        fl, hh, ww = 0, net_h, net_w
        sizes = []
        for cin, cout, k, s, _ in conv_specs():
            sizes.append((cin, cout, k, s))
        self.flops_per_frame = _yolo_flops(net_h, net_w)
        # HBM bytes per pixel the lane pixel stages have to move (bench.py): read BGR 3, write the non-maximum-suppressed
        # magnitudes 1, read them for the hysteresis pass 1 (the resolve / compaction passes only touch the ROI box: +0.3).
        # SURVEY 8d's 7*W*H per frame assumed the blurred image goes out to memory and back; the fused front end keeps it
        # in registers, so the chain as a whole is priced on 7*W*H and this stage on what it really needs
        self.lane_pixel_bytes_per_px = 5
        self.lane_pixel_kernels = ("front_pack (gray+blur+hist+Sobel+NMS) + thresholds + ccl_tile + ccl_border + "
                                   "finalize_fast + compact_box")
        self.frame_idx = 0
        self._lanes_pending = False
        self._tail_deferred = False
        self.stream.synchronize()

    @property
    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def enqueue_generate(self, stream0=0):
        nat.check(self.L.av_synth_frames(self.ctx.handle, self._s, self.S, self.h, self.w, stream0, self.frame_idx,
                                         nat.ptr(self.frames)))
        self.frame_idx += 1

    def enqueue_detect(self):
        nat.check(self.L.av_yolo_forward(self.yolo._h, self._s, nat.ptr(self.frames), 0.25, 0.7, self.max_det,
                                         nat.ptr(self.det_n), nat.ptr(self.det_box), nat.ptr(self.det_conf),
                                         nat.ptr(self.det_cls)))

    def enqueue_lanes(self, stream=None, stages=0):
        """stages 0: the whole chain; 2: pixel stages only (edge points left in the workspace); 16: Hough + fit of
        what the last pixel-stage call left there."""
        nat.check(self.L.av_lane_detect(self.ctx.handle, stream or self._s, C.byref(self.lcfg), self.S, self.h, self.w,
                                        nat.ptr(self.frames), None, nat.ptr(self.ws), nat.ptr(self.lane_state),
                                        nat.ptr(self.poly), nat.ptr(self.pts), nat.ptr(self.info), nat.ptr(self.conf), stages))

    def step(self, sync=False):
        """generate; fork{lanes} || {detect}; join.  The detector and the lane chain only share the frames: the
        lane chain (mostly the sequential per-frame PPHT, one CU per frame) runs beside the convolutions."""
        h = self.ctx.handle
        self.enqueue_generate()
        nat.check(self.L.av_fork(h, self._s))
        self.enqueue_lanes(self.ctx.side_stream)
        self.enqueue_detect()
        nat.check(self.L.av_join(h, self._s))
        if sync:
            self.stream.synchronize()

    def tune_streams(self, candidates=8, steps=4):
        """Pick the main stream the step overlaps best on.  A step is four chains on four streams (detector main chain, Detect
        head's class branch, lane chain, deferred detector tail); the HIP runtime serves a process's streams from a few hardware
        queues per priority class, assigned by the process's whole stream history, and WHICH queues the four chains got decides
        how they overlap -- measured on MI355X / ROCm 7.2: the same 64-camera step takes 1.55 ms or 2.9 ms (tools/c3seq.py: fresh
        process 1.55; after any HotLoop was closed 2.9; with four more highest-priority streams created first 1.55 again).  The
        runtime offers no way to ask for a queue, so the loop measures: `steps` pipelined steps on each of `candidates`
        highest-priority streams (PyTorch hands its pool out round-robin), keeps the fastest.  Results do not depend on the
        stream; the generator's frame counter and the lanes' EMA state are put back afterwards.  -> ms per step of every candidate."""
        import time
        state0, idx0 = self.lane_state.clone(), self.frame_idx
        deferred = bool(self._tail_deferred)
        self.synchronize()
        tried = []
        for _ in range(max(1, candidates)):
            st = torch.cuda.Stream(device=self.dev, priority=-1)
            self.stream = st
            self._lanes_pending = False
            for _ in range(2):
                self.step_deferred()
            self.flush_lanes()
            self.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step_deferred()
            self.flush_lanes()
            self.synchronize()
            tried.append(((time.perf_counter() - t0) / steps * 1e3, st))
        best = min(tried, key=lambda x: x[0])
        self.stream = best[1]
        self._lanes_pending = False
        self.lane_state.copy_(state0)
        self.frame_idx = idx0
        self.synchronize()
        self.defer_detector_tail(deferred)
        return [round(t, 4) for t, _ in tried]

    def defer_detector_tail(self, enable=True):
        """Throughput mode of the detector: decode + sort + NMS of step k beside the convolutions of step k+1
        (av_yolo_defer_tail); det_* are complete after flush_lanes() / join_detector_tail()."""
        if enable and self.yolo.precision == "fp32":
            raise RuntimeError("the float32 detector mode has no deferred tail")
        nat.check(self.L.av_yolo_defer_tail(self.yolo._h, 1 if enable else 0))
        self._tail_deferred = bool(enable)

    def join_detector_tail(self):
        nat.check(self.L.av_yolo_join_tail(self.yolo._h, self._s))

    def step_deferred(self):
        """Throughput variant of step(): the Hough + fit half of a frame's lane chain is enqueued one step late, ahead
        of the next frame's pixel stages on the side stream.  The sharded PPHT holds 158 KB of LDS on every CU it
        runs on, which keeps the LDS-tiled convolutions off those CUs; one step late it runs beside the detector's
        preprocess + stem + first stride-2 convolution, which use no LDS.  Same kernels, same per-stream order, same
        results; the lane outputs (poly / pts / info / conf) describe the PREVIOUS frame until flush_lanes()."""
        h = self.ctx.handle
        self.enqueue_generate()
        nat.check(self.L.av_fork(h, self._s))
        if self._lanes_pending:
            self.enqueue_lanes(self.ctx.side_stream, stages=16)
        self.enqueue_lanes(self.ctx.side_stream, stages=2)
        self._lanes_pending = True
        self.enqueue_detect()
        nat.check(self.L.av_join(h, self._s))

    def flush_lanes(self):
        """Hough + fit of the last frame step_deferred() left pending."""
        if self._lanes_pending:
            self.enqueue_lanes(stages=16)
            self._lanes_pending = False
        self.join_detector_tail()

    def synchronize(self):
        self.stream.synchronize()


def _yolo_flops(H, W):
    """2*MACs of the YOLOv8n graph at input H x W (per frame), walking the same layer list as the library."""
    from .perception.yolo import conv_specs
    specs = conv_specs()
    # spatial size of every conv's OUTPUT, in execution order
    def c2f(n):
        return [0] * (2 + 2 * n)
    div = []
    div += [2, 4] + [4] * 4 + [8] + [8] * 6 + [16] + [16] * 6 + [32] + [32] * 4 + [32, 32]     # backbone + SPPF
    div += [16] * 4 + [8] * 4 + [16] + [16] * 4 + [32] + [32] * 4                                # head
    for d in (8, 16, 32):
        div += [d] * 6
    assert len(div) == len(specs)
    fl = 0
    for (cin, cout, k, s, _), d in zip(specs, div):
        fl += 2 * cin * cout * k * k * (H // d) * (W // d)
    return fl
