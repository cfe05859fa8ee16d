"""ctypes binding of libavhot.so (include/avhot.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device
is visible when a context is requested, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavhot.so")

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)
vp = C.c_void_p


class TrackerCfg(C.Structure):
    _fields_ = [("iou_threshold", C.c_double), ("max_age", C.c_int32), ("min_hits", C.c_int32),
                ("trajectory_length", C.c_int32)]


class KfCfg(C.Structure):
    _fields_ = [("dt", C.c_double), ("process_noise", C.c_double), ("measurement_noise", C.c_double)]


class PlannerCfg(C.Structure):
    _fields_ = [("planning_horizon", C.c_double), ("dt", C.c_double), ("num_samples", C.c_int32),
                ("reserved", C.c_int32), ("w_lateral", C.c_double), ("w_velocity", C.c_double),
                ("w_acceleration", C.c_double), ("w_curvature", C.c_double)]


class LaneCfg(C.Structure):
    _fields_ = [("hough_threshold", C.c_int32), ("min_line_length", C.c_int32), ("max_line_gap", C.c_int32),
                ("max_segments", C.c_int32), ("smoothing_factor", C.c_double)]


# numpy structured dtype mirroring av_track_row (64 bytes)
TRACK_ROW_FIELDS = [("id", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"), ("cls", "<i4"),
                    ("age", "<i4"), ("hits", "<i4"), ("misses", "<i4"), ("slot", "<i4"), ("hist_len", "<i4"),
                    ("flags", "<i4"), ("conf", "<f8"), ("vx", "<f4"), ("vy", "<f4")]
TRACK_ROW_BYTES = 64
TRACKER_HDR_BYTES = 64
KF_STATE_DOUBLES = 48
MANEUVER_STATE_DOUBLES = 32
MANEUVER_ROW_FIELDS = [("lateral", "<i4"), ("longitudinal", "<i4"), ("turning", "<i4"), ("reserved", "<i4"),
                       ("lateral_confidence", "<f8"), ("longitudinal_confidence", "<f8"), ("turning_confidence", "<f8"),
                       ("speed_kmh", "<f8"), ("acceleration", "<f8"), ("yaw_rate_deg", "<f8"), ("timestamp", "<f8")]
MANEUVER_ROW_BYTES = 72
INTERACTION_ROW_FIELDS = [("type", "<i4"), ("risk", "<i4"), ("agent_id", "<i4"), ("cls", "<i4"), ("confidence", "<f8"),
                          ("distance", "<f8"), ("relative_speed", "<f8"), ("ttc", "<f8")]
INTERACTION_ROW_BYTES = 48
INTERACTION_SUMMARY_FIELDS = [("agent_count", "<i4"), ("pedestrian_count", "<i4"), ("cyclist_count", "<i4"),
                              ("vehicle_count", "<i4"), ("n_interactions", "<i4"), ("primary_type", "<i4"),
                              ("overall_risk", "<i4"), ("primary_row", "<i4"), ("closest_distance", "<f8"),
                              ("min_ttc", "<f8"), ("timestamp", "<f8")]
INTERACTION_SUMMARY_BYTES = 56


class BevCfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixels_per_meter", C.c_double), ("x_min", C.c_double),
                ("x_max", C.c_double), ("y_min", C.c_double), ("y_max", C.c_double), ("n_candidates", C.c_int32),
                ("reserved", C.c_int32)]


class InteractionCfg(C.Structure):
    _fields_ = [("frame_h", C.c_int32), ("frame_w", C.c_int32), ("class_kind", C.c_int32 * 16)]
PRIM_FIELDS = [("type", "<i4"), ("x0", "<i4"), ("y0", "<i4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"),
               ("x3", "<i4"), ("y3", "<i4"), ("p", "<i4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"), ("reserved", "<i4")]
PRIM_BYTES = 48
PRIM_RECT, PRIM_SEG, PRIM_QUAD, PRIM_DISC, PRIM_RING, PRIM_GLYPH, PRIM_BLEND_RECT, PRIM_POLY_BLEND = 1, 2, 3, 4, 5, 6, 7, 8
VSTATE_DOUBLES = 12
WP_DOUBLES = 6

# (name, restype, argtypes); everything returns int status except the three noted
_SIGS = [
    ("av_version", C.c_int, []),
    ("av_last_error_string", C.c_char_p, []),
    ("av_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("av_ctx_create", C.c_int, [C.c_int, C.POINTER(vp)]),
    ("av_ctx_destroy", C.c_int, [vp]),
    ("av_ctx_device", C.c_int, [vp, C.POINTER(C.c_int)]),
    ("av_side_stream", C.c_int, [vp, C.POINTER(vp)]),
    ("av_fork", C.c_int, [vp, vp]),
    ("av_join", C.c_int, [vp, vp]),
    ("av_graph_begin", C.c_int, [vp, vp]),
    ("av_graph_end", C.c_int, [vp, vp, C.POINTER(C.c_int)]),
    ("av_graph_launch", C.c_int, [vp, C.c_int, vp]),
    ("av_graph_destroy", C.c_int, [vp, C.c_int]),
    ("av_event_create", C.c_int, [C.POINTER(vp)]),
    ("av_event_destroy", C.c_int, [vp]),
    ("av_event_record", C.c_int, [vp, vp]),
    ("av_event_elapsed_ms", C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    ("av_stream_sync", C.c_int, [vp]),
    ("av_stream_sync_spin", C.c_int, [vp]),
    ("av_host_alloc", C.c_int, [C.POINTER(vp), C.c_size_t]),
    ("av_host_free", C.c_int, [vp]),
    ("av_copy_h2d", C.c_int, [vp, vp, C.c_size_t, vp]),
    ("av_copy_d2h", C.c_int, [vp, vp, C.c_size_t, vp, C.c_int]),
    ("av_simdet_generate", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    ("av_tracker_state_bytes", C.c_size_t, [C.c_int, C.c_int]),
    ("av_tracker_reset", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    ("av_tracker_update", C.c_int, [vp, vp, C.POINTER(TrackerCfg), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                    C.c_int, vp, vp, vp, vp]),
    ("av_wire_table_bytes", C.c_size_t, [C.c_int]),
    ("av_pack_tracks", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    ("av_comm_unique_id", C.c_int, [vp]),
    ("av_comm_create", C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
    ("av_comm_destroy", C.c_int, [vp]),
    ("av_allgather_tracks", C.c_int, [vp, vp, vp, vp, vp, C.c_size_t]),
    ("av_kf_reset", C.c_int, [vp, vp, C.c_int, vp]),
    ("av_kf_step", C.c_int, [vp, vp, C.POINTER(KfCfg), C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    ("av_planner_configure", C.c_int, [vp, C.POINTER(PlannerCfg)]),
    ("av_planner_dims", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("av_planner_plan", C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]),
    ("av_hot_step", C.c_int, [vp, vp, C.POINTER(TrackerCfg), C.POINTER(KfCfg)] + [C.c_int] * 5 + [vp] * 17 + [vp, C.c_int, C.c_int]),
    ("av_hot_steps_seq", C.c_int, [vp, C.c_int, vp, C.POINTER(TrackerCfg), C.POINTER(KfCfg)] + [C.c_int] * 5 + [vp] * 7 +
     [C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    ("av_hot_step_seq", C.c_int, [vp, vp, C.POINTER(TrackerCfg), C.POINTER(KfCfg)] + [C.c_int] * 5 + [vp] * 17 +
     [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    ("av_planner_generate", C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp]),
    ("av_planner_evaluate", C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]),
    ("av_lane_workspace_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("av_lane_workspace_init", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("av_lane_workspace_view", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_size_t)]),
    ("av_raster_draw", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int]),
    ("av_bev_prim_cap", C.c_int, [vp, C.c_int, C.c_int]),
    ("av_bev_build", C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]),
    ("av_i420_to_bgr", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    ("av_resize_into", C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("av_synth_frames", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("av_maneuver_reset", C.c_int, [vp, vp, C.c_int, vp]),
    ("av_interaction_state_bytes", C.c_size_t, [C.c_int]),
    ("av_interaction_reset", C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    ("av_interaction_detect", C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    ("av_maneuver_detect", C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    ("av_yolo_param_count", C.c_size_t, []),
    ("av_yolo_create", C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(vp)]),
    ("av_yolo_create_ex", C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(vp)]),
    ("av_yolo_destroy", C.c_int, [vp]),
    ("av_yolo_dims", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("av_yolo_forward", C.c_int, [vp, vp, vp, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp]),
    ("av_yolo_keep_logits", C.c_int, [vp, C.c_int]),
    ("av_yolo_defer_tail", C.c_int, [vp, C.c_int]),
    ("av_yolo_join_tail", C.c_int, [vp, vp]),
    ("av_yolo_tensor", C.c_int, [vp, C.c_int, C.POINTER(vp)] + [C.POINTER(C.c_int)] * 5),
    ("av_lane_detect", C.c_int, [vp, vp, C.POINTER(LaneCfg), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                 vp, C.c_int]),
]

# entry points added by later translation units; bound when present in the header list below
_OPTIONAL_SIGS = []

_lib = None


def declared_symbols():
    """Every function name include/avhot.h declares (parsed from the header text)."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "avhot.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(av_[a-z0-9_]+)\s*\(", txt)))


def _preload_hip_runtime():
    """One HIP runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  libavhot.so
    NEEDs libamdhip64.so.7 too; if the system copy under /opt/rocm were mapped first, torch would
    later map its bundled copy as a second runtime, and the two do not share devices, streams or
    allocations.  Mapping torch's copy first makes the dynamic loader satisfy our NEEDED entry
    with it (SONAME match).
    """
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def lib():
    """The loaded library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libavhot.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "-- this package has no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, res, args in _SIGS + _OPTIONAL_SIGS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def register(sigs):
    """Used by lane/yolo bindings to add their signatures before first use."""
    global _lib
    _OPTIONAL_SIGS.extend(sigs)
    if _lib is not None:
        for name, res, args in sigs:
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args


def step_flag_ints(n_streams):
    """AV_STEP_FLAG_INTS(n_streams) of include/avhot.h: int32 words of the overlapped steps' sequence flags."""
    return 65 * n_streams + 32 + 64


def step_i32(v):
    """A step number (kept modulo 2^32) as the signed int the C ABI takes."""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


class StepSet(C.Structure):
    """av_step_set: the per-step buffers of one parity (av_hot_steps_seq)."""
    _fields_ = [(k, C.c_void_p) for k in ("det_n", "det_box", "det_cls", "det_conf", "snap", "snap_n", "det2trk", "z", "vstate",
                                          "plan_state", "waypoints", "cost", "order")]


def check(rc):
    if rc != 0:
        raise RuntimeError("libavhot: %s (code %d)" % (lib().av_last_error_string().decode(), rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


class Context:
    """Owns one av_ctx on one device."""

    def __init__(self, device=0):
        L = lib()
        h = vp()
        check(L.av_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        s = vp()
        check(L.av_side_stream(h, C.byref(s)))
        self.side_stream = s

    def close(self):
        if getattr(self, "handle", None):
            lib().av_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (the per-frame drop-in classes share it)."""
    c = _default_ctx.get(device)
    if c is None or c.handle is None:
        c = Context(device)
        _default_ctx[device] = c
    return c


def stream_handle(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream) as c_void_p."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)
