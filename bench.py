#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the perception -> tracking -> planning hot loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 it is launched under torch.distributed.run, one rank per GPU (RCCL).

Headline (default) = BASELINE config 4 AS WORDED ("64 streams ... one graph launch per time-step"), the per-GPU
share of config 5: 64 concurrent synthetic streams per GPU batched through detect (simulated) -> track -> Kalman step -> plan,
a step = ONE TIME-STEP (one frame of every stream) = one launch of the one-kernel step av_hot_step (launched, not replayed:
replaying a one-kernel graph costs 5 us more per step than launching the kernel), all inputs resident in HBM.
value = frames/s over all ranks.  A reported "step" is timed over `inner_reps` back-to-back repetitions so that every figure
rests on >= 0.2 s of device time (`--steps` is what the driver passes; ms_per_step = time / (steps x inner_reps)).

At N = 1 the same JSON line carries, under "also", the other single-GPU configurations measured in the same process:
  config4_window256  the throughput form of config 4: 256-frame windows of every stream per step, one hipGraph replay per
               window (needs the window's 256 future measurement frames: an offline / replay mode, not a live loop)
  config3      YOLO-mode detector (MFMA convs) + Canny/Hough lane detector on device-generated 1280x720 frames
  config2      1 stream, simulated detection (latency-bound: one dependent chain)
  config2_w1   config 2 with window 1
each as a compact summary (value, ms_per_step, dtype, roofline of the stage that dominates that configuration's step, CPU
baseline); the printed line stays under 8 KB.  The full records -- every configuration's per-stage kernel list (isolated
HIP-event times on the launch stream, share of their sum, what bounds the stage, achieved vs peak) -- go to the side file
named in "detail_file" (default gpurun_out/bench_detail.json; a builder-run copy is committed as profiles/rNN_bench_kernels.json).
`traffic` values are PMC counter results read from profiles/*_pmc.json; each of those files carries the hash of the kernel
source it was measured on, and a value whose source has changed since is printed as null ("traffic_stale": true).
`cpu_baseline` legs are timed BEFORE the GPU is initialised (the nproc leg starts child processes).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 / f16 MFMA ~2.5 PF
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: float32-operand MFMA (v_mfma_f32_16x16x4_f32) = the float32 vector rate
N_CUS = 256
CPU_SHARE = 16                 # host cores that go with one GPU of an 8-GPU node (process / thread pools are sized to it)
PKG = "multimodal_autonomous_driving_perception_and_planning_amd"
METRIC = "end-to-end frames/sec (1280x720 synthetic)"

# algorithmic HBM bytes per stream-frame
TRACKER_BYTES = 2 * 48 * 48 + 7 * 24        # read+write hot table rows (T=48) + detections (D=7)  = 4776
KF_BYTES = 768
SIMDET_BYTES = 228                           # det_n + 7 x (box 16 + cls 4 + conf 8) written
LANE_BYTES_PER_PX = 7


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config4", choices=["config2", "config3", "config4"])
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default by workload)")
    ap.add_argument("--window", type=int, default=None, help="frames per stream per step")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None, help="replay the step as a captured hipGraph")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--no-also", action="store_true", help="headline configuration only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--native-allgather", action="store_true",
                    help="N>1: gather with the library's own av_allgather_tracks (RCCL communicator made from a broadcast "
                         "ncclUniqueId) instead of torch.distributed.all_gather_into_tensor")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"], help="config3: detector arithmetic")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N>1, window 1: time-steps per all-gather (the per-frame tables of k steps in one message)")
    ap.add_argument("--overlap", type=int, default=4, choices=[1, 2, 3, 4],
                    help="window 1: D > 1 = up to D consecutive time-steps in flight (HotLoop(overlap=D): still one launch per step, "
                         "ordered per stream and role by device-side counters); 1 = each launch after the previous one has drained")
    ap.add_argument("--min-seconds", type=float, default=0.2, help="least device time behind every reported figure")
    ap.add_argument("--gather", default="window-end", choices=["window-end", "per-frame"],
                    help="N>1: all-gather the end-of-window table (cheap) or every frame's table of the window")
    ap.add_argument("--no-defer", action="store_true", help="config3: whole lane chain inside its own step")
    ap.add_argument("--no-tune", action="store_true", help="keep the loops' first streams (no tune_streams(): which hardware queues a "
                                                           "process's streams get depends on its stream history)")
    ap.add_argument("--taggers", action="store_true",
                    help="also run the maneuver and interaction taggers (SURVEY 8f-3) in every step")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="side file for the full per-configuration records (kernel lists)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (oracle = checker; timed here as the reported baseline only).  No GPU call before or inside.
# ---------------------------------------------------------------------------------------------------------------
_CHILD = ("import sys; sys.path.insert(0, %r); from oracle.harness_ref import time_cpu_loop; "
          "print(time_cpu_loop(int(sys.argv[1]), warmup=5))" % ROOT)


def cpu_hot_loop(seconds):
    """NumPy oracle of the simulated-detection loop: 1 process x 1 thread, then nproc independent processes."""
    from oracle.harness_ref import time_cpu_loop
    fps0 = time_cpu_loop(40, warmup=3)
    n = max(60, int(fps0 * seconds))
    fps = time_cpu_loop(n, warmup=5)
    out = {"value": round(fps, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the single-stream 1280x720 simulated-detection loop (detect+track+KF+plan), "
                     "NumPy oracle, 1 process x 1 thread" % n}
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    P = max(1, min(ncpu, CPU_SHARE))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", _CHILD, str(n)], stdout=subprocess.PIPE, env=env) for _ in range(P)]
    ok = 0
    for p in procs:
        o, _ = p.communicate()
        ok += 1 if p.returncode == 0 and o.strip() else 0
    el = time.perf_counter() - t0
    if ok == P:
        out["nproc"] = {"value": round(P * (n + 5) / el, 2), "unit": "frames/s", "cores": P, "cpus_visible": ncpu,
                        "sample": "%d independent single-stream processes x %d frames, wall time incl. interpreter start" % (P, n + 5)}
    return out


def cpu_pixel_path(seconds):
    """Config 3 on the host: C lane oracle (1 thread) + PyTorch-CPU fp32 YOLOv8n-topology forward, decode and NMS."""
    import numpy as np
    import torch
    from oracle import lane_ref, yolo_ref
    frames = [lane_ref.synthetic_frame(720, 1280, s, 0) for s in range(4)]
    lr = lane_ref.LaneRef()
    lr.detect(frames[0])
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds * 0.4 or n < 4:
        lr.detect(frames[n % 4])
        n += 1
    lane_ms = (time.perf_counter() - t0) / n * 1e3
    net = yolo_ref.build_model(yolo_ref.random_params(0))
    torch.set_num_threads(max(1, min(torch.get_num_threads(), CPU_SHARE)))
    thr = torch.get_num_threads()

    def one(fr):
        with torch.no_grad():
            x = torch.from_numpy(yolo_ref.preprocess(fr))[None]
            f = net.features(x)
            b, c, k = yolo_ref.decode(f["head"])
            keep = yolo_ref.nms(b, c, k)
            return yolo_ref.scale_boxes(b[keep], 720, 1280)
    one(frames[0])
    t0, m = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds * 0.6 or m < 3:
        one(frames[m % 4])
        m += 1
    yolo_ms = (time.perf_counter() - t0) / m * 1e3
    return {"value": round(1e3 / (lane_ms + yolo_ms), 3), "unit": "frames/s", "cores": thr, "kind": "port",
            "lane_ms_per_frame": round(lane_ms, 2), "yolo_ms_per_frame": round(yolo_ms, 2),
            "sample": "%d frames through oracle/c/lane_ref.c (1 thread) + %d frames through oracle/yolo_ref.py "
                      "(PyTorch-CPU fp32, %d intra-op threads), 1280x720 synthetic" % (n, m, thr)}


# ---------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------
class Events:
    def __init__(self, L, nat, n):
        self.L, self.nat = L, nat
        self.ev = [[C.c_void_p(), C.c_void_p()] for _ in range(n)]
        for e in self.ev:
            nat.check(L.av_event_create(C.byref(e[0])))
            nat.check(L.av_event_create(C.byref(e[1])))

    def avg_ms(self):
        ms, tot = C.c_float(), 0.0
        for e in self.ev:
            self.nat.check(self.L.av_event_elapsed_ms(e[0], e[1], C.byref(ms)))
            tot += ms.value
        return tot / len(self.ev)

    def close(self):
        for e in self.ev:
            self.L.av_event_destroy(e[0])
            self.L.av_event_destroy(e[1])


def time_stage(L, nat, stream, fn, reps, sync):
    """Average HIP-event time of fn() alone on `stream` (the stream it is launched on)."""
    fn()
    sync()
    ev = Events(L, nat, reps)
    for e in ev.ev:
        nat.check(L.av_event_record(e[0], stream))
        fn()
        nat.check(L.av_event_record(e[1], stream))
    sync()
    ms = ev.avg_ms()
    ev.close()
    return ms


def _pmc_load(name):
    """A counter profile under profiles/, or None when it is missing or no longer describes the kernel in the tree: every
    profile carries the sha256 (first 16 hex digits) of the kernel source file it was measured on (tools/profiles_from_db.py)."""
    import hashlib
    p = os.path.join(ROOT, "profiles", name)
    try:
        j = json.load(open(p))
        src = os.path.join(ROOT, PKG, j["kernel_source"])
        if hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] != j["kernel_source_sha256_16"]:
            return None
        return j
    except Exception:
        return None


def pmc_traffic(name, key, scale):
    j = _pmc_load(name)
    return int(j[key] * scale) if j is not None and key in j else None


def pmc_value(name, key):
    j = _pmc_load(name)
    return j.get(key) if j is not None else None


def pmc_stamp(name):
    j = _pmc_load(name)
    return {"file": "profiles/" + name, "commit": j.get("measured_at_commit")} if j is not None else {"file": "profiles/" + name, "stale": True}


def finish_kernel_list(ks):
    tot = sum(k["avg_ms"] for k in ks)
    for k in ks:
        k["share_of_stage_sum"] = round(k["avg_ms"] / tot, 4)
        k["avg_ms"] = round(k["avg_ms"], 5)
    return ks


def pick_reps(est_step_s, steps, min_s, world):
    """Inner repetitions per reported step so that the timed region lasts >= min_s; every rank uses rank 0's choice."""
    import torch
    import torch.distributed as dist
    r = max(1, int(1.4 * min_s / max(est_step_s * steps, 1e-9) + 0.999))     # (the probe includes a drain: it overestimates)
    if world > 1:
        t = torch.tensor([r], dtype=torch.int64, device="cuda")
        dist.broadcast(t, 0)
        r = int(t.item())
    return r


def max_over_ranks(el, world):
    import torch
    import torch.distributed as dist
    if world <= 1:
        return el, [round(el * 1e3, 4)]
    t = torch.tensor([el], dtype=torch.float64, device="cuda")
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    per = [float(x.item()) for x in allt]
    return max(per), [round(p * 1e3, 4) for p in per]


# ---------------------------------------------------------------------------------------------------------------
# configs 2 / 4: simulated detection -> tracker || Kalman -> planner
# ---------------------------------------------------------------------------------------------------------------
class ChainBroken(RuntimeError):
    """An overlapped HotLoop reported a step that waited in vain for its predecessor (on this rank or another)."""


def run_hot_loop(a, world, rank, local, name, S, W, graph, steps, warmup, stage_reps=10, overlap=None):
    import numpy as np
    import torch
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.distributed import TrackTableExchange
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop

    # window 1, launched (not graph-replayed): consecutive steps overlapped unless asked otherwise
    ov = (a.overlap if overlap is None else overlap) if (W == 1 and not graph and not a.taggers) else 1
    def any_rank(flag):                  # the ranks decide together: a collective follows
        if world > 1:
            t = torch.tensor([int(bool(flag))], device=torch.device("cuda", local))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return bool(int(t.item()))
        return bool(flag)

    # Overlapped launches need kernels of different streams to run concurrently; under a tool that runs one kernel at a time in an
    # order of its own (rocprofv3 --pmc) a step waits in vain for its predecessor and the loop reports it (fault word): ChainBroken
    # on every rank, and the caller measures with serial launches instead.
    loop, tuned, why = None, None, None
    try:
        loop = HotLoop(n_streams=S, window=W, device=local, keep_waypoints=True, keep_snapshots=True, overlap=ov)
        # (which hardware queues the D streams got decides how the launches overlap)
        tuned = loop.tune_streams() if (loop.overlap > 1 and not a.no_tune) else None
    except (RuntimeError, ValueError) as e:
        if ov == 1:
            raise
        why = str(e)
    if ov > 1 and any_rank(why is not None):
        raise ChainBroken(why or "another rank's overlapped loop reported a broken chain")

    def chain_ok():
        if loop.overlap > 1 and any_rank(int(loop.seq_flags[64 * S].item()) != 0):
            raise ChainBroken("a step waited in vain for its predecessor")
    ov = loop.overlap
    L = nat.lib()
    g0 = rank * S                                      # global stream ids of this rank
    loop.reset(frame_offsets=[(g0 + s) * 17 for s in range(S)])   # offset s*17
    # synthetic ego measurements: one seeded sequence per stream, re-used every window (input data only)
    zlen = max(W, 64)
    z = np.stack([np.asarray(generate_ego_motion(zlen, seed=g0 + s), np.float64)[:W] for s in range(S)])
    if ov > 1:
        loop.load_measurements(z, all_sets=True)
    else:
        loop.load_measurements(z)
    xchg = None
    if world > 1 and not a.no_allgather:
        # window 1: every step IS a frame, so the per-frame gather (config 5's wording) is the default there -- and with the
        # one-launch step the wire tables come out of the step kernel itself (no pack launch)
        # window 1: the tables of `--gather-every` consecutive time-steps travel in one all-gather (every frame's table is gathered,
        # k steps at a time: a 13-us step cannot wait for a collective of its own)
        bucket = max(1, a.gather_every) if (W == 1 and loop.fused_step) else 1
        xchg = TrackTableExchange(loop, world, rank, per_frame=(a.gather == "per-frame" or W == 1), native=a.native_allgather, bucket=bucket)
    h, s = loop.ctx.handle, loop._s
    cross = xchg is not None or a.taggers     # somebody on the main stream reads the tracker's tables every step
    # overlapped loop: the launch loop of `unit` consecutive time-steps is one library call (av_hot_steps_seq; a Python call per
    # launch costs 8 us, the device needs 7.4) -- with a gather, one bucket of `--gather-every` steps and its all-gather
    unit = (xchg.bucket if xchg is not None else max(1, a.gather_every)) if ov > 1 else 1

    def one_step():
        if ov > 1:
            if xchg is not None:
                xchg.step_bucket()
            else:
                loop.enqueue_steps(unit)
            return
        if xchg is not None:
            xchg.begin_step()                 # (one-launch step: this step writes its wire tables into the exchange's send buffer)
        if graph:
            loop.step(graph=True)             # fork{detect; track} || {kf; plan}; join -- inside the graph (window 1: one kernel)
        elif loop.fused_step:
            loop.enqueue_step()               # window 1: one launch
        else:
            # side stream: detect -> track; main: kf -> plan.  When the main stream reads the tracker's tables every
            # step (exchange / interaction tagger) the side stream must first wait for the previous step's readers
            if cross:
                nat.check(L.av_fork(h, s))
            loop.enqueue_detect(loop.ctx.side_stream)
            loop.enqueue_track(loop.ctx.side_stream)
            loop.enqueue_kf()
            loop.enqueue_plan()
            if a.taggers:
                loop.enqueue_maneuver()
            if cross:
                nat.check(L.av_join(h, s))
        if a.taggers:
            if graph:
                loop.enqueue_maneuver()
            loop.enqueue_interactions()
        if xchg is not None:
            xchg.exchange()

    def drain():
        if not graph and not loop.fused_step:
            nat.check(L.av_join(h, s))       # main stream waits for the side stream's tail
        if loop.overlap > 1:
            loop.synchronize(check=False)    # (the fault word is looked at by all ranks together: chain_ok)
        else:
            loop.synchronize()
        if xchg is not None:
            xchg.flush()                      # (a partially filled bucket)
            xchg.synchronize()

    for _ in range(max(warmup, 1)):
        one_step()
    drain()
    torch.cuda.synchronize()
    chain_ok()
    # a probe of the step's duration (untimed) sizes the inner repetitions: every reported step = `reps` back-to-back steps
    tp = time.perf_counter()
    for _ in range(8):
        one_step()
    drain()
    torch.cuda.synchronize()
    reps = pick_reps((time.perf_counter() - tp) / (8 * unit), steps, a.min_seconds, world)
    calls = -(-steps * reps // unit)          # whole calls of `unit` time-steps
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(calls):
        one_step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    el, per_rank_ms = max_over_ranks(el, world)
    nsteps = calls * unit
    chain_ok()
    # the tracker's sticky overflow flag: a truncated table would silently drop births (parity lost)
    hdr, _, _ = loop.tracker_tables()
    if int(np.abs(hdr[:, 3]).max()) != 0:
        raise RuntimeError("tracker table overflow (hdr[3] != 0): tcap %d too small for this run" % loop.tcap)
    t_inflight = None
    if ov > 1:
        # a launch's own duration while `ov` are in flight (HIP events around every launch on the stream it runs on, the loop
        # stepping launch by launch): what rocprofv3 --kernel-trace reports for hot_step_kernel on this command.  It contains the
        # waits for the predecessor; the step rate above is `ov` of these at a time.
        for _ in range(64):
            loop.enqueue_step()
        loop.synchronize(check=False)
        ev = Events(L, nat, 256)
        for e in ev.ev:
            st = C.c_void_p(loop._pstreams[loop._seq % ov].cuda_stream)
            nat.check(L.av_event_record(e[0], st))
            loop.enqueue_step()
            nat.check(L.av_event_record(e[1], st))
        loop.synchronize(check=False)
        t_inflight = ev.avg_ms()
        ev.close()
        chain_ok()
        # the stage kernels and the step kernel alone are timed on a serial loop of the same shape
        del loop
        if xchg is not None:
            xchg.loop = None
        loop = HotLoop(n_streams=S, window=W, device=local, keep_waypoints=True, keep_snapshots=True)
        loop.reset(frame_offsets=[(g0 + s_) * 17 for s_ in range(S)])
        loop.load_measurements(z)
        for _ in range(200):
            loop.enqueue_step()
        loop.synchronize()
        # the kernel launched serially, back to back (device-bound: the host needs 8 us per launch, the kernel 13): its duration as
        # rocprofv3 --kernel-trace reports it for `--overlap 1` (profiles/r04_bench_config4_serial_kernel_stats.csv)
        ta = time.perf_counter()
        for _ in range(3000):
            loop.enqueue_step()
        loop.synchronize()
        t_alone = (time.perf_counter() - ta) / 3000 * 1e3
        s = loop._s

    # per-stage kernel times, each stage alone on the launch stream
    F = S * W
    sync = loop.synchronize
    ks = []
    t_det = time_stage(L, nat, s, loop.enqueue_detect, stage_reps, sync)
    t_trk = time_stage(L, nat, s, loop.enqueue_track, stage_reps, sync)
    t_kf = time_stage(L, nat, s, loop.enqueue_kf, stage_reps, sync)
    t_pl = time_stage(L, nat, s, loop.enqueue_plan, stage_reps, sync)
    pb = loop.planner_bytes_per_state() * F

    def hbm(nbytes, ms):
        return round(nbytes / (ms * 1e-3) / 1e9, 2)
    ks.append({"kernel": "tracker_kernel", "stage": "track", "branch": "side", "avg_ms": t_trk, "bound": "latency",
               "why": "one workgroup per stream walks its frames in order (frame t+1 needs frame t's table)",
               "cus_occupied": min(S, N_CUS), "us_per_frame_per_stream": round(t_trk * 1e3 / W, 3),
               "bytes_per_launch": TRACKER_BYTES * F, "achieved": hbm(TRACKER_BYTES * F, t_trk), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(hbm(TRACKER_BYTES * F, t_trk) / HBM_PEAK_GBS, 5),
               "traffic": pmc_traffic("tracker_pmc.json", "hbm_bytes_per_frame", F)})
    ks.append({"kernel": "planner_wave_kernel", "stage": "plan", "branch": "main", "avg_ms": t_pl,
               "bound": "hbm" if F >= 4096 else "latency", "bytes_per_launch": pb, "achieved": hbm(pb, t_pl),
               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm(pb, t_pl) / HBM_PEAK_GBS, 4),
               "traffic": pmc_traffic("planner_pmc.json", "hbm_bytes_per_state", F)})
    ks.append({"kernel": "kf_axis_kernel", "stage": "kf", "branch": "main", "avg_ms": t_kf, "bound": "latency",
               "why": "one wave per stream, sequential predict/update chain", "cus_occupied": min((S + 3) // 4, N_CUS),
               "bytes_per_launch": KF_BYTES * F, "achieved": hbm(KF_BYTES * F, t_kf), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(hbm(KF_BYTES * F, t_kf) / HBM_PEAK_GBS, 5)})
    ks.append({"kernel": "simdet_kernel", "stage": "detect", "branch": "side", "avg_ms": t_det, "bound": "latency",
               "bytes_per_launch": SIMDET_BYTES * F, "achieved": hbm(SIMDET_BYTES * F, t_det), "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": round(hbm(SIMDET_BYTES * F, t_det) / HBM_PEAK_GBS, 5)})
    side, main = t_det + t_trk, t_kf + t_pl
    stage_kernels = None
    if loop.fused_step:
        # the step IS one kernel: it is the roofline entry; the four stage kernels it replaces are kept beside it for comparison
        # (an empty event bracket on the same stream is timed too and taken off: two event packets around a 13-us kernel are a fifth of
        # the reading, and rocprofv3's kernel duration -- which the committed summaries carry -- has none)
        t_empty = time_stage(L, nat, s, lambda: None, 4 * stage_reps, sync)
        t_fused = max(time_stage(L, nat, s, loop.enqueue_step_fused, 4 * stage_reps, sync) - t_empty, 0.0)
        if t_inflight is not None:
            t_inflight = max(t_inflight - t_empty, 0.0)
        stage_kernels = finish_kernel_list(ks)
        ks = [{"kernel": "hot_step_kernel", "stage": "detect + track || Kalman + plan (one launch, role-split workgroups)", "branch": "main",
               "avg_ms": t_fused, "bound": "latency",
               "why": "one frame per stream and launch: per stream a detector row + tracker frame in one workgroup, Kalman step + 21 "
                      "trajectories in another; %d of %d CUs, microseconds of dependent work each" % (min(2 * S, N_CUS), N_CUS),
               "cus_occupied": min(2 * S, N_CUS), "bytes_per_launch": (TRACKER_BYTES + KF_BYTES + SIMDET_BYTES) * F + pb,
               "achieved": hbm((TRACKER_BYTES + KF_BYTES + SIMDET_BYTES) * F + pb, t_fused), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(hbm((TRACKER_BYTES + KF_BYTES + SIMDET_BYTES) * F + pb, t_fused) / HBM_PEAK_GBS, 5),
               "event_bracket_ms_taken_off": round(t_empty, 5),
               "traffic": pmc_traffic("hot_step_pmc.json", "hbm_bytes_per_stream_step", F)}]
    if ov > 1:
        # roofline of the kernel = its bytes over its own duration (launched serially); in flight a launch is stretched by its waits
        k0 = ks[0]
        k0["avg_ms_event_bracket_alone"] = k0["avg_ms"]
        k0["avg_ms"] = t_alone
        k0["achieved"] = hbm(k0["bytes_per_launch"], t_alone)
        k0["frac"] = round(k0["achieved"] / HBM_PEAK_GBS, 5)
        k0["avg_ms_in_flight_event_bracket"] = round(t_inflight, 5)
        k0["launches_in_flight"] = ov
        k0["launch_interval_ms"] = round(el / nsteps * 1e3, 6)
        k0["achieved_at_launch_interval"] = hbm(k0["bytes_per_launch"], el / nsteps * 1e3)      # what the launches in flight move together
        k0["why"] += ("; consecutive launches overlap (%d in flight, ordered per stream and role by device-side counters): a launch " % ov +
                      "alone takes avg_ms, in flight longer (its waits), and one completes every launch_interval_ms")
    dom = max(ks, key=lambda k: k["avg_ms"])
    roof = {k: dom[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "launches_in_flight",
                                "launch_interval_ms", "achieved_at_launch_interval", "avg_ms_in_flight_event_bracket") if k in dom}
    roof["avg_launch_ms"] = round(dom["avg_ms"], 5)
    roof["traffic"] = dom.get("traffic")
    if dom.get("traffic") is None and dom["kernel"] in ("tracker_kernel", "planner_wave_kernel", "hot_step_kernel"):
        roof["traffic_stale"] = True         # the counter profile was measured on another version of the kernel's source
    if dom["bound"] == "latency":
        roof["cus_occupied"] = dom.get("cus_occupied")
        roof["note"] = ("dominant kernel is a per-stream sequential chain on %d of %d CUs; its HBM fraction is reported for "
                        "completeness%s" % (dom.get("cus_occupied", 0), N_CUS,
                                            "" if loop.fused_step else ", the HBM-bound kernel of this step is the planner (see kernels)"))
    planner_k = next((k for k in (stage_kernels or ks) if k["kernel"] == "planner_wave_kernel"), None)
    if planner_k is not None and not loop.fused_step:
        roof["hbm_bound_kernel_beside_it"] = {k: planner_k[k] for k in ("kernel", "avg_ms", "achieved", "frac", "traffic", "bytes_per_launch")}
    step_bytes = (TRACKER_BYTES + KF_BYTES + SIMDET_BYTES) * F + pb
    out = {"metric": METRIC, "value": round(F * nsteps * world / el, 1), "unit": "frames/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "inner_reps": reps, "timed_s": round(el, 4),
           "ms_per_step": round(el / nsteps * 1e3, 6), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s: %d stream(s)/GPU x %d-frame window per step, 1280x720, simulated detection + IoU "
                                  "tracker + 6-state KF + 21-candidate planner%s" % (name, S, W, ", one hipGraph replay per step" if graph else
                                                              ((", one launch per step, up to %d consecutive steps in flight" % ov if ov > 1 else ", one launch per step") if W == 1 else "")),
                      "streams_per_gpu": S, "window": W, "graph": bool(graph), "taggers": bool(a.taggers),
                      "allgather_track_tables": (("per-frame" if xchg.per_frame else "window-end") if xchg is not None else False),
                      "allgather_impl": (("av_allgather_tracks (RCCL)" if xchg.native else "torch.distributed") if xchg is not None else None),
                      "fused_step": bool(loop.fused_step), "overlapped_steps": ov, "steps_per_library_call": unit,
                      "stream_sets_tried_us_per_step": tuned,
                      "parallelism": "stream-sharded x%d" % world},
           "roofline": roof, "kernels": finish_kernel_list(ks),
           "step": {"critical_branch": "side (detect+track)" if side > main else "main (kf+plan)",
                    "side_branch_ms": round(side, 5), "main_branch_ms": round(main, 5),
                    "algorithmic_bytes": step_bytes,
                    "hbm_frac_whole_step": round(step_bytes / (el / nsteps) / 1e9 / HBM_PEAK_GBS, 4)},
           "ranks_seen": (dist.get_world_size() if world > 1 else 1), "per_rank_ms": per_rank_ms}
    if xchg is not None:
        out["config"]["allgather_bytes_per_rank_per_step"] = xchg.bytes_per_step
        out["config"]["allgather_every_steps"] = xchg.bucket
    if stage_kernels is not None:
        out["stage_kernels_replaced"] = stage_kernels
    out["traffic_profiles"] = [pmc_stamp("hot_step_pmc.json")] if loop.fused_step else [pmc_stamp("tracker_pmc.json"), pmc_stamp("planner_pmc.json")]
    if W == 1:
        out["us_per_time_step"] = round(el / nsteps * 1e6, 3)
    del loop, xchg
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# config 3: YOLO-mode detector + lane detector on device-generated frames
# ---------------------------------------------------------------------------------------------------------------
def run_config3(a, world, rank, local, S, steps, warmup, stage_reps=6, precision="fp16"):
    import torch
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import PerceptionLoop
    loop = PerceptionLoop(n_streams=S, device=local, precision=precision)
    L = nat.lib()
    s, side = loop._s, loop.ctx.side_stream
    fp32 = precision == "fp32"
    mfma_peak = MFMA_F32_PEAK_TFLOPS if fp32 else MFMA_PEAK_TFLOPS
    if not a.no_defer and not fp32:
        loop.defer_detector_tail(True)       # decode + sort + NMS of step k beside the convolutions of step k+1
    # which hardware queues the runtime gave the step's four streams decides how its chains overlap (1.55 or 2.9 ms per step for the
    # same work, depending on the process's stream history): the loop measures a few candidate main streams and keeps the best
    tuned = loop.tune_streams() if (not a.no_defer and not a.no_tune) else None
    s = loop._s

    def one_step():
        # PerceptionLoop.step_deferred(): the lane chain runs beside the detector on the side stream (both only read
        # the frames), its Hough + fit half one step late so that it meets the detector's LDS-free first kernels
        loop.enqueue_generate(stream0=rank * S)
        nat.check(L.av_fork(loop.ctx.handle, s))
        if not a.no_defer:
            if loop._lanes_pending:
                loop.enqueue_lanes(side, stages=16)
            loop.enqueue_lanes(side, stages=2)
            loop._lanes_pending = True
        else:
            loop.enqueue_lanes(side)
        loop.enqueue_detect()
        nat.check(L.av_join(loop.ctx.handle, s))

    for _ in range(max(warmup, 1)):
        one_step()
    loop.synchronize()
    torch.cuda.synchronize()
    tp = time.perf_counter()
    for _ in range(4):
        one_step()
    loop.synchronize()
    torch.cuda.synchronize()
    reps = pick_reps((time.perf_counter() - tp) / 4, steps, a.min_seconds, world)
    nsteps = steps * reps
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        one_step()
    loop.flush_lanes()            # the last frame's Hough half is flushed inside the timed region
    loop.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    el, per_rank_ms = max_over_ranks(el, world)

    if not fp32:
        loop.defer_detector_tail(False)      # the per-stage times below are of whole stages
    sync = loop.synchronize
    t_gen = time_stage(L, nat, s, loop.enqueue_generate, stage_reps, sync)
    t_yolo = time_stage(L, nat, s, loop.enqueue_detect, stage_reps, sync)
    t_pix = time_stage(L, nat, s, lambda: loop.enqueue_lanes(stages=2), stage_reps, sync)
    t_hough = time_stage(L, nat, s, lambda: loop.enqueue_lanes(stages=16), stage_reps, sync)
    px = S * loop.h * loop.w
    fl = loop.flops_per_frame * S
    tf = fl / (t_yolo * 1e-3) / 1e12
    pix_bytes = loop.lane_pixel_bytes_per_px * px
    ks = [
        {"kernel": ("yolo forward, float32 operands: preprocess + 63 conv_f32_kernel launches (v_mfma_f32_16x16x4_f32, one generic "
                    "implicit-GEMM kernel, no fusion) + 3 max-pools + 2 upsamples + decode + radix sort + NMS") if fp32 else
                   ("yolo forward: front_fused_kernel (letterbox + stem + layer 1) + c2f16_fused_kernel (layer 2) + c2f32_head/tail "
                    "kernels (layers 4, 15) + 47 conv launches (conv3x3_ws_kernel / conv1x1_ws_kernel / conv_lds_kernel / conv_mfma_kernel; "
                    "decode in the head's last convolutions) + sppf + upsample + radix sort + NMS"), "stage": "detect", "branch": "main",
         "avg_ms": t_yolo, "bound": "mfma",
         "flops_per_launch": int(fl), "achieved": round(tf, 2), "peak": mfma_peak, "unit": "TFLOP/s",
         "frac": round(tf / mfma_peak, 4),
         "traffic": None if fp32 else pmc_traffic("yolo_hbm_pmc.json", "hbm_bytes_per_forward", S / 64.0)},
        {"kernel": "lane pixel stages: " + loop.lane_pixel_kernels, "stage": "lane (pixels)", "branch": "side",
         "avg_ms": t_pix, "bound": "hbm", "bytes_per_launch": pix_bytes,
         "achieved": round(pix_bytes / (t_pix * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(pix_bytes / (t_pix * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "traffic": pmc_traffic("lane_pmc.json", "pixel_stage_hbm_bytes_per_px", px)},
        {"kernel": "lane Hough + fit: hough_prep + houghp_shard (+ houghp_fast / houghp_kernel fallbacks) + lane_fit",
         "stage": "lane (Hough)", "branch": "side", "avg_ms": t_hough, "bound": "latency",
         "why": "sequential probabilistic Hough per frame (cv::RNG order), 4 single-wave workgroups per frame",
         "cus_occupied": min(4 * S, N_CUS), "bytes_per_launch": px, "achieved": round(px / (t_hough * 1e-3) / 1e9, 1),
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(px / (t_hough * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        {"kernel": "synth_rows_kernel", "stage": "generate", "branch": "main", "avg_ms": t_gen, "bound": "hbm",
         "bytes_per_launch": 3 * px, "achieved": round(3 * px / (t_gen * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(3 * px / (t_gen * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
    ]
    lane_ms = t_pix + t_hough
    lane_total = LANE_BYTES_PER_PX * px
    dom = max(ks, key=lambda k: k["avg_ms"])
    roof = {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
            "unit": dom["unit"], "frac": dom["frac"], "traffic": dom.get("traffic"),
            "avg_launch_ms": round(dom["avg_ms"], 5)}
    if "flops_per_launch" in dom:
        roof["flops_per_launch"] = dom["flops_per_launch"]
        if not fp32:
            roof["mfma_busy_pmc_percent"] = pmc_value("yolo_mfma_pmc.json", "overall_mfma_busy_percent")
    out = {"metric": METRIC, "value": round(S * nsteps * world / el, 1), "unit": "frames/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "inner_reps": reps, "timed_s": round(el, 4),
           "ms_per_step": round(el / nsteps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": loop.yolo.precision, "data": "synthetic (generated on device)",
           "config": {"workload": "config3: %d camera streams/GPU, one 1280x720 frame of each per step, YOLO-mode detector "
                                  "(random-init YOLOv8n topology, letterbox 384x640%s) + Canny/Hough lane detector" % (S, ", float32 operands: the reference's own precision" if fp32 else ", IEEE-half operands"),
                      "streams_per_gpu": S, "parallelism": "stream-sharded x%d" % world,
                      "main_stream_candidates_ms": tuned},
           "roofline": roof, "kernels": finish_kernel_list(ks),
           "lane_chain": {"avg_ms": round(lane_ms, 5), "bytes_per_launch": lane_total,
                          "achieved": round(lane_total / (lane_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(lane_total / (lane_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "SURVEY 8d: 7*W*H algorithmic bytes per frame over pixel stages + Hough + fit"},
           "ranks_seen": (dist.get_world_size() if world > 1 else 1), "per_rank_ms": per_rank_ms}
    out["traffic_profiles"] = [pmc_stamp("lane_pmc.json"), pmc_stamp("yolo_mfma_pmc.json"), pmc_stamp("yolo_hbm_pmc.json")]
    del loop
    torch.cuda.empty_cache()
    return out


def run_per_frame_classes(local, frames=220, warm=20):
    """The reference's own call pattern (demo.py:97-120): the five drop-in classes, ONE 1280x720 frame per call, results on
    the host after every call.  Mean / median ms per frame and per stage, with and without the lane detector."""
    import numpy as np
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion, synthetic_frame
    from src.perception import LaneDetector, ObjectDetector
    from src.planning import MotionPlanner
    from src.state_estimation import VehicleStateEstimator
    from src.tracking import MultiObjectTracker
    det, lane, trk, est, pl = (ObjectDetector(mode="simulated", device=local), LaneDetector(device=local),
                               MultiObjectTracker(device=local), VehicleStateEstimator(device=local), MotionPlanner(device=local))
    ego = generate_ego_motion(frames + warm)
    imgs = [synthetic_frame(720, 1280, 0, f) for f in range(8)]
    names = ("detect", "lane", "track", "kf", "plan")
    T = {k: [] for k in names}
    for i in range(frames + warm):
        fr = imgs[i % 8]
        t0 = time.perf_counter()
        d = det.detect(fr)
        t1 = time.perf_counter()
        lane.detect(fr)
        t2 = time.perf_counter()
        trk.update(d)
        t3 = time.perf_counter()
        st = est.step(np.array(ego[i]))
        t4 = time.perf_counter()
        pl.plan((st.x, st.y, st.heading, st.speed))
        t5 = time.perf_counter()
        if i >= warm:
            for k, v in zip(names, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                T[k].append(v * 1e3)
    T = {k: np.array(v) for k, v in T.items()}
    # the YOLO-mode detector through the same class surface (random-init YOLOv8n topology): one pinned upload, one mapped result read
    ydet = ObjectDetector(mode="yolo", model_path="random:0", device=local)
    ty = []
    for i in range(60):
        t0 = time.perf_counter()
        ydet.detect(imgs[i % 8])
        if i >= 10:
            ty.append((time.perf_counter() - t0) * 1e3)
    no_lane = T["detect"] + T["track"] + T["kf"] + T["plan"]
    tot = no_lane + T["lane"]
    return {"frames": frames, "what": "ObjectDetector(simulated) -> LaneDetector -> MultiObjectTracker -> VehicleStateEstimator -> "
                                      "MotionPlanner, one 1280x720 host frame per call, host-visible results after every call",
            "ms_per_frame_median": round(float(np.median(tot)), 4), "ms_per_frame_mean": round(float(tot.mean()), 4),
            "ms_per_frame_without_lanes_median": round(float(np.median(no_lane)), 4),
            "ms_per_frame_without_lanes_mean": round(float(no_lane.mean()), 4),
            "frames_per_s_median": round(1e3 / float(np.median(tot)), 1),
            "stage_ms_median": {k: round(float(np.median(v)), 4) for k, v in T.items()},
            "yolo_mode_detect_ms_median": round(float(np.median(ty)), 4),
            "yolo_mode_what": "ObjectDetector(mode='yolo', model_path='random:0').detect(frame): upload + 1-image forward + sort + NMS + "
                              "host-visible Detection list"}


def run_bev_panels(local, S=64, W=16, reps=10):
    """the BEV panel of every stream built and painted on the device from the step's tables."""
    import numpy as np
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    loop = HotLoop(n_streams=S, window=W, device=local)
    loop.reset(frame_offsets=[s * 17 for s in range(S)])
    loop.load_measurements(np.stack([np.asarray(generate_ego_motion(W, seed=s), np.float64) for s in range(S)]))
    for _ in range(4):
        loop.step(sync=True)
    ms = time_stage(nat.lib(), nat, loop._s, loop.enqueue_bev, reps, loop.synchronize)
    px = S * 600 * 600
    return {"panels": S, "size": "600x600", "avg_ms": round(ms, 4), "panels_per_s": round(S / ms * 1e3, 1),
            "kernels": "copy of the road image + bev_build_kernel + raster_kernel",
            "bytes_per_launch": 12 * px, "achieved": round(12 * px / (ms * 1e-3) / 1e9, 1), "unit": "GB/s",
            "note": "12 B/px: the road image read and written by the copy, read and written again by the rasteriser; the per-tile "
                    "walk over the primitive list, not memory, sets the time"}


def _short(txt, n=140):
    return txt if len(txt) <= n else txt[:n - 3] + "..."


def compact_summary(v):
    """One `also` entry as the driver keeps it: value, time per step, dtype, the dominant stage's roofline, the CPU baseline."""
    if "value" not in v:                                  # per_frame_classes / bev_panels: already small
        return {k: x for k, x in v.items() if not isinstance(x, str)}
    out = {"value": v["value"], "unit": v.get("unit"), "ms_per_step": v.get("ms_per_step"), "dtype": v.get("dtype"),
           "workload": _short(v.get("config", {}).get("workload", ""), 90)}
    if "us_per_time_step" in v:
        out["us_per_time_step"] = v["us_per_time_step"]
    r = v.get("roofline", {})
    out["roofline"] = {k: (_short(r[k], 60) if isinstance(r[k], str) else r[k])
                       for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms",
                                 "cus_occupied", "mfma_busy_pmc_percent") if k in r}
    hb = r.get("hbm_bound_kernel_beside_it")
    if hb and hb.get("kernel") != r.get("kernel"):
        out["roofline"]["hbm_bound_kernel_beside_it"] = {k: hb[k] for k in ("kernel", "avg_ms", "frac", "traffic") if k in hb}
    if "lane_chain" in v:
        out["lane_chain"] = {k: v["lane_chain"][k] for k in ("avg_ms", "achieved", "frac", "unit")}
        out["stage_ms"] = {k["stage"]: k["avg_ms"] for k in v.get("kernels", [])}
    c = v.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "cores", "kind", "gpu_over_cpu", "gpu_over_cpu_nproc") if k in c}
        if "nproc" in c:
            out["cpu_baseline"]["nproc"] = {k: c["nproc"][k] for k in ("value", "cores")}
    return out


def compact_line(head):
    """The printed line: the headline in full except its long texts, `also` as compact summaries."""
    line = {k: v for k, v in head.items() if k not in ("also", "kernels", "stage_kernels_replaced")}
    line["kernels"] = [{k: (_short(x, 100) if isinstance(x, str) else x) for k, x in kk.items() if k not in ("why", "peak", "unit", "branch")}
                       for kk in head.get("kernels", [])]
    if "roofline" in line and "note" in line["roofline"]:
        line["roofline"] = dict(line["roofline"], note=_short(line["roofline"]["note"], 120))
    if "cpu_baseline" in line and "sample" in line["cpu_baseline"]:
        line["cpu_baseline"] = dict(line["cpu_baseline"], sample=_short(line["cpu_baseline"]["sample"], 200))
    if "also" in head:
        line["also"] = {k: compact_summary(v) for k, v in head["also"].items()}
    return line


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus,
              file=sys.stderr)
        sys.exit(2)

    # CPU legs first: nothing has touched the GPU yet, so child processes are safe to start
    cpu_hot = cpu_pix = None
    if world == 1 and not a.no_cpu_baseline:
        cpu_hot = cpu_hot_loop(a.cpu_seconds)
        if not a.no_also or a.workload == "config3":
            cpu_pix = cpu_pixel_path(a.cpu_seconds)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def with_cpu(line, cpu):
        if cpu is not None:
            line["cpu_baseline"] = dict(cpu)
            line["cpu_baseline"]["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
            if "nproc" in cpu:
                line["cpu_baseline"]["gpu_over_cpu_nproc"] = round(line["value"] / cpu["nproc"]["value"], 1)
        return line

    def hot(name, S, W, graph, steps, warmup, overlap=None):
        try:
            return run_hot_loop(a, world, rank, local, name, S, W, graph, steps, warmup, overlap=overlap)
        except ChainBroken as e:
            # (raised on every rank together) the same measurement with serial launches
            r = run_hot_loop(a, world, rank, local, name, S, W, graph, steps, warmup, overlap=1)
            r["config"]["overlap_fallback"] = str(e)[:200]
            return r

    if a.workload == "config3":
        head = with_cpu(run_config3(a, world, rank, local, a.streams or 64, a.steps, a.warmup, precision=a.precision), cpu_pix)
    elif a.workload == "config2":
        g = bool(a.graph) if a.graph is not None else False
        head = with_cpu(hot("config2", a.streams or 1, a.window or 131072, g, a.steps, a.warmup), cpu_hot)
    else:
        W = a.window or 1                                              # as worded: one launch per time-step
        g = (W != 1) if a.graph is None else bool(a.graph)             # window 1 is one kernel: launched, not replayed
        head = with_cpu(hot("config4" if W == 1 else "config4 (%d-frame windows)" % W, a.streams or 64, W, g, a.steps, a.warmup), cpu_hot)

    if world == 1 and not a.no_also and a.workload == "config4":
        also = {}
        if (a.window or 1) == 1:
            if head["config"].get("overlapped_steps", 1) > 1:
                # the same steps with every launch waiting for the previous one to drain (round 3's form of the headline)
                also["config4_serial_launches"] = with_cpu(hot("config4", a.streams or 64, 1, False, 20, 5, overlap=1), cpu_hot)
            also["config4_window256"] = with_cpu(hot("config4, throughput form: 256-frame windows, one hipGraph replay per window",
                                                     a.streams or 64, 256, True, 20, 5), cpu_hot)
        else:
            also["config4_w1"] = with_cpu(hot("config4 (window 1)", a.streams or 64, 1, False, 20, 5), cpu_hot)
        also["config3"] = with_cpu(run_config3(a, world, rank, local, 64, 20, 5), cpu_pix)
        # the same configuration in the reference's own arithmetic (ultralytics runs float32): what the half-precision figure above is to be read against
        also["config3_fp32"] = with_cpu(run_config3(a, world, rank, local, 64, 5, 2, stage_reps=3, precision="fp32"), cpu_pix)
        also["config2"] = with_cpu(hot("config2", 1, 32768, False, 4, 1), cpu_hot)
        also["config2_w1"] = with_cpu(hot("config2 (window 1)", 1, 1, False, 20, 5), cpu_hot)
        also["config4_256streams"] = with_cpu(hot("config4 scaled to 256 streams (not a BASELINE config: shows the "
                                                  "HBM-bound regime once every CU has a tracker stream)", 256, 256, True, 10, 3), cpu_hot)
        also["per_frame_classes"] = run_per_frame_classes(local)
        also["bev_panels"] = run_bev_panels(local)
        for v in also.values():          # the headline's fixed keys stay on the headline only
            for k in ("metric", "higher_is_better", "scaling", "vs_baseline", "n_gpus", "ranks_seen", "per_rank_ms"):
                v.pop(k, None)
        head["also"] = also

    if rank == 0:
        # full records (kernel lists) -> side file; the printed line carries compact per-configuration summaries (< 8 KB)
        detail = json.loads(json.dumps(head))
        try:
            os.makedirs(os.path.dirname(a.detail_out), exist_ok=True)
            with open(a.detail_out, "w") as f:
                json.dump(detail, f, indent=1)
            head["detail_file"] = os.path.relpath(a.detail_out, ROOT)
        except OSError:
            head["detail_file"] = None
        print(json.dumps(compact_line(head)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
