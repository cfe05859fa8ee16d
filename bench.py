#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the perception -> tracking -> planning hot loop on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 it is launched under torch.distributed.run, one rank per GPU (RCCL).

A "step" is one window of the hot loop over the rank's streams: detect (simulated) -> track ->
Kalman step -> plan for `window` consecutive frames of each of `streams` video streams, all inputs
(ego measurements, detector counters) already resident in HBM.  value = frames/s over all ranks.

Workloads (BASELINE.json configs):
  config2  1 stream  per GPU, 1280x720, simulated detection + IoU tracker + KF + 21-candidate planner
  config4  64 streams per GPU, same stages (the per-GPU share of config5's 512 streams on 8 GPUs)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4"])
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (default by workload)")
    ap.add_argument("--window", type=int, default=None, help="frames per stream per step")
    ap.add_argument("--graph", action="store_true", help="replay the step as a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-defer", action="store_true", help="config3: whole lane chain inside its own step (no one-step-late Hough half)")
    ap.add_argument("--taggers", action="store_true",
                    help="also run the maneuver and interaction taggers (SURVEY 8f-3) in every step")
    return ap.parse_args()


def cpu_baseline(seconds):
    """The CPU oracle ("port" of the reference's NumPy path) on a bounded sample of the same loop."""
    from oracle.harness_ref import time_cpu_loop
    fps0 = time_cpu_loop(40, warmup=3)
    n = max(60, int(fps0 * seconds))
    fps = time_cpu_loop(n, warmup=5)
    return {"value": round(fps, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the single-stream 1280x720 simulated-detection loop (detect+track+KF+plan), "
                      "NumPy oracle, 1 thread" % n}


MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PF


def bench_config3(a, world, rank, local):
    """YOLO-mode detector (random-init YOLOv8n topology, bf16 MFMA convs) + lane detector on frames generated
    on the device.  step = one frame of each of `streams` cameras."""
    import torch
    import torch.distributed as dist
    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import PerceptionLoop
    S = a.streams or 64
    loop = PerceptionLoop(n_streams=S, device=local)
    L = nat.lib()
    evs = [[C.c_void_p(), C.c_void_p()] for _ in range(a.steps)]
    for e in evs:
        nat.check(L.av_event_create(C.byref(e[0])))
        nat.check(L.av_event_create(C.byref(e[1])))
    for _ in range(a.warmup):
        loop.step()
    loop.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        # PerceptionLoop.step_deferred() with events around the detector: the lane chain runs beside it on the side
        # stream (both only read the frames), its Hough + fit half one step late so that it meets the detector's
        # LDS-free first kernels; the last frame's half is flushed inside the timed region
        loop.enqueue_generate(stream0=rank * S)
        nat.check(L.av_fork(loop.ctx.handle, loop._s))
        if not a.no_defer:
            if loop._lanes_pending:
                loop.enqueue_lanes(loop.ctx.side_stream, stages=16)
            loop.enqueue_lanes(loop.ctx.side_stream, stages=2)
            loop._lanes_pending = True
        else:
            loop.enqueue_lanes(loop.ctx.side_stream)
        nat.check(L.av_event_record(evs[k][0], loop._s))
        loop.enqueue_detect()
        nat.check(L.av_event_record(evs[k][1], loop._s))
        nat.check(L.av_join(loop.ctx.handle, loop._s))
    loop.flush_lanes()
    loop.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms = C.c_float()
    tot = 0.0
    for e in evs:
        nat.check(L.av_event_elapsed_ms(e[0], e[1], C.byref(ms)))
        tot += ms.value
    det_ms = tot / a.steps
    tfl = loop.flops_per_frame * S / (det_ms * 1e-3) / 1e12
    if rank == 0:
        out = {"metric": "end-to-end frames/sec (1280x720 synthetic)", "value": round(S * a.steps * world / el, 1),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic (generated on device)",
               "config": {"workload": "config3: %d camera streams/GPU, 1280x720, YOLO-mode detector (random-init YOLOv8n "
                                      "topology, letterbox 384x640) + Canny/Hough lane detector" % S,
                          "streams_per_gpu": S, "parallelism": "stream-sharded x%d" % world},
               "roofline": {"bound": "mfma", "kernel": "conv_lds_kernel / conv_mfma_kernel (63 launches per forward; timed with preprocess, decode and NMS, beside the lane chain)",
                            "achieved": round(tfl, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tfl / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                            "flops_per_launch": int(loop.flops_per_frame * S), "avg_launch_ms": round(det_ms, 4)}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % a.gpus,
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
    if a.workload == "config3":
        return bench_config3(a, world, rank, local)
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
    from multimodal_autonomous_driving_perception_and_planning_amd.distributed import TrackTableExchange
    from multimodal_autonomous_driving_perception_and_planning_amd.harness import generate_ego_motion

    S = a.streams or (1 if a.workload == "config2" else 64)
    W = a.window or (131072 if a.workload == "config2" else 256)
    loop = HotLoop(n_streams=S, window=W, device=local, keep_waypoints=True, keep_snapshots=True)
    L = nat.lib()
    g0 = rank * S                                      # global stream ids of this rank
    loop.reset(frame_offsets=[(g0 + s) * 17 for s in range(S)])   # offset s*17
    # synthetic ego measurements: one seeded sequence per stream, re-used every window (input data only)
    z = np.stack([np.asarray(generate_ego_motion(W, seed=g0 + s), np.float64) for s in range(S)])
    loop.load_measurements(z)
    xchg = None
    if world > 1 and not a.no_allgather:
        xchg = TrackTableExchange(loop, world, rank)

    evs = [[C.c_void_p(), C.c_void_p()] for _ in range(a.steps)]
    for e in evs:
        nat.check(L.av_event_create(C.byref(e[0])))
        nat.check(L.av_event_create(C.byref(e[1])))

    def one_step(k, timed):
        if a.graph:
            loop.step(graph=True)
        else:
            # side stream: detect -> track (stream order alone keeps steps apart there); main: kf -> plan; join.
            # No fork is needed outside graph capture: nothing on the side stream depends on the main one.
            h, s = loop.ctx.handle, loop._s
            loop.enqueue_detect(loop.ctx.side_stream)
            loop.enqueue_track(loop.ctx.side_stream)
            loop.enqueue_kf()
            if timed:
                nat.check(L.av_event_record(evs[k][0], s))
            loop.enqueue_plan()
            if timed:
                nat.check(L.av_event_record(evs[k][1], s))
            if a.taggers:
                loop.enqueue_maneuver()                   # consumes the Kalman output (main stream)
            # the two chains share no buffer, so they only have to meet when somebody reads across them: per step
            # for the track-table exchange or the interaction tagger, otherwise once before the final synchronisation
            if xchg is not None or a.taggers:
                nat.check(L.av_join(h, s))
            if a.taggers:
                loop.enqueue_interactions()               # consumes the tracker's tables and the Kalman output
        if xchg is not None:
            xchg.exchange()

    def drain():
        if not a.graph:
            nat.check(L.av_join(loop.ctx.handle, loop._s))       # main stream waits for the side stream's tail
        loop.synchronize()

    for k in range(a.warmup):
        one_step(k, False)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        one_step(k, True)
    drain()
    if xchg is not None:
        xchg.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # dominant-by-bytes kernel: the planner (51.7 KB written per frame vs 4.8 KB tracker, 0.8 KB KF)
    if a.graph:
        for k in range(a.steps):      # graph nodes cannot be bracketed; time the same launches directly
            nat.check(L.av_event_record(evs[k][0], loop._s))
            loop.enqueue_plan()
            nat.check(L.av_event_record(evs[k][1], loop._s))
        loop.synchronize()
    ms = C.c_float()
    tot = 0.0
    for e in evs:
        nat.check(L.av_event_elapsed_ms(e[0], e[1], C.byref(ms)))
        tot += ms.value
    plan_ms = tot / a.steps
    plan_bytes = loop.planner_bytes_per_state() * S * W
    ach = plan_bytes / (plan_ms * 1e-3) / 1e9

    if rank == 0:
        frames = S * W * a.steps * world
        # HBM bytes per launch from the PMC counters (WRITE_SIZE + corrected FETCH_SIZE), collected with
        # rocprofv3 in separate passes and committed under profiles/ (per start state, scaled to this launch)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "planner_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = int(json.load(open(pmc))["hbm_bytes_per_state"] * S * W)
            except Exception:
                traffic = None
        out = {
            "metric": "end-to-end frames/sec (1280x720 synthetic)", "value": round(frames / el, 1),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d stream(s)/GPU x %d-frame window, 1280x720, simulated detection + IoU "
                                   "tracker + 6-state KF + 21-candidate planner" % (a.workload, S, W),
                       "streams_per_gpu": S, "window": W, "graph": bool(a.graph), "taggers": bool(a.taggers),
                       "allgather_track_tables": bool(xchg is not None), "parallelism": "stream-sharded x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "planner_wave_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": plan_bytes, "avg_launch_ms": round(plan_ms, 5)},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
