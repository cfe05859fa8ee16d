import sys, os, json
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench, torch
a = bench.parse()
if os.environ.get("KEEP"):
    torch.cuda.empty_cache = lambda: None        # freed blocks stay in torch's pool instead of going back to the driver
def c3(tag):
    nd = int(os.environ.get("DUMMY", "0"))
    global _dummies
    _dummies = []
    for _ in range(nd):                      # shift the hardware-queue id the loop's main stream will get
        st = torch.cuda.Stream(priority=-1)
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda")
        st.synchronize(); _dummies.append(st)
    if os.environ.get("ADDR"):
        import ctypes as C
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import PerceptionLoop
        lp = PerceptionLoop(n_streams=64)
        ptrs = {"frames": lp.frames.data_ptr(), "ws": lp.ws.data_ptr()}
        for tid in (0, 1, 2, 4, 6, 15, 21):
            pp, H, W, Cc, cs, co = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
            lp.L.av_yolo_tensor(lp.yolo._h, tid, C.byref(pp), C.byref(H), C.byref(W), C.byref(Cc), C.byref(cs), C.byref(co))
            ptrs["t%d" % tid] = pp.value
        print(tag, {k: (hex(v), "align %d KB" % ((v & -v) >> 10)) for k, v in ptrs.items()}, flush=True)
        del lp
    r = bench.run_config3(a, 1, 0, 0, 64, 20, 5)
    print(tag, "config3 ms/step", r["ms_per_step"], "reps", r["inner_reps"], flush=True)
which = os.environ.get("SEQ", "c3")
for step in which.split(","):
    if step == "c3": c3("after " + which)
    elif step == "ctx":
        from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
        keep = nat.Context(0); print("ctx made", flush=True)
    elif step == "stream":
        keep2 = torch.cuda.Stream(); print("stream made", flush=True)
    elif step == "hstream":
        keep3 = torch.cuda.Stream(priority=-1); print("high-priority stream made", flush=True)
    elif step == "loop0":
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        keep4 = HotLoop(n_streams=64, window=1); print("HotLoop made", flush=True)
    elif step == "loop1":
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        keep5 = HotLoop(n_streams=64, window=1); keep5.step(sync=True); print("HotLoop stepped", flush=True)
    elif step in ("many", "det", "trk", "kf", "plan", "w256trk"):
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        lp = HotLoop(n_streams=64, window=256 if step == "w256trk" else 1)
        if step == "many":
            for _ in range(20000): lp.step()
        elif step == "det": lp.enqueue_detect()
        elif step in ("trk", "w256trk"): lp.enqueue_detect(); lp.enqueue_track()
        elif step == "kf": lp.enqueue_kf()
        elif step == "plan": lp.enqueue_kf(); lp.enqueue_plan()
        lp.synchronize(); print(step, "done", flush=True)
        keepx = lp
    elif step in ("loopdel", "ctxdel", "graphdel"):
        import gc
        from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        if step == "ctxdel":
            c = nat.Context(0); c.close(); del c
        else:
            lp = HotLoop(n_streams=64, window=256 if step == "graphdel" else 1)
            lp.step(graph=(step == "graphdel"), sync=True)
            del lp
        gc.collect(); torch.cuda.empty_cache(); print(step, "done", flush=True)
    elif step in ("e1", "e2", "e3", "e4", "e5"):
        import gc, ctypes as C
        from multimodal_autonomous_driving_perception_and_planning_amd import _native as nat
        from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import HotLoop
        if step == "e1":                     # loop deleted, its context kept alive
            c = nat.Context(0); lp = HotLoop(n_streams=64, window=1, ctx=c); lp.step(sync=True); del lp; keep_c = c
        elif step == "e2":                   # small tensors only
            lp = HotLoop(n_streams=64, window=1, keep_waypoints=False); lp.step(sync=True); del lp
        elif step == "e3":                   # context + planner table, closed
            c = nat.Context(0); lp = HotLoop(n_streams=1, window=1, ctx=c); del lp; c.close()
        elif step == "e4":                   # torch tensors of the loop's sizes, freed
            ts = [torch.zeros(n, dtype=torch.uint8, device="cuda") for n in (256, 64 * 8 * 16, 64 * 107000, 64 * 21 * 51 * 48, 64 * 4096, 4096)]
            torch.cuda.synchronize(); del ts
        elif step == "e5":                   # loop deleted but never stepped
            lp = HotLoop(n_streams=64, window=1); lp.synchronize(); del lp
        gc.collect(); torch.cuda.empty_cache(); print(step, "done", flush=True)
    elif step == "w1": r = bench.run_hot_loop(a, 1, 0, 0, "w1", 64, 1, False, 20, 5); print("w1", r["ms_per_step"], flush=True)
    elif step == "w256": r = bench.run_hot_loop(a, 1, 0, 0, "w256", 64, 256, True, 20, 5); print("w256", r["ms_per_step"], flush=True)
    elif step == "w256ng": r = bench.run_hot_loop(a, 1, 0, 0, "w256ng", 64, 256, False, 20, 5); print("w256 no graph", r["ms_per_step"], flush=True)
if os.environ.get("TRACE_C3"):
    # how the step time of a config-3 loop evolves after a light-load phase (groups of 10 steps, synchronised)
    import time
    from multimodal_autonomous_driving_perception_and_planning_amd.pipeline import PerceptionLoop
    loop = PerceptionLoop(n_streams=64)
    loop.defer_detector_tail(True)
    out = []
    for g in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            loop.step_deferred()
        loop.synchronize(); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 100, 3))
    print("ms/step per group of 10:", out, flush=True)
