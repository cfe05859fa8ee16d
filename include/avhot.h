/*
 * avhot.h -- C ABI of libavhot.so: the MI355X (gfx950) hot path of the
 * per-frame detect -> lane -> track -> Kalman -> plan loop.
 *
 * The reference (bhavyageethika/multimodal_autonomous_driving_perception_and_planning)
 * is pure Python and has no FFI; the drop-in boundary is its five classes
 * (SURVEY.md section 8b).  This header is what those classes' replacements
 * bind through ctypes; each entry point names the reference method it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - Every function returns 0 (AV_OK) or a negative AV_E* code and never
 *     throws; av_last_error_string() describes the last failure on this thread.
 *   - All data pointers are DEVICE pointers unless the parameter is documented
 *     as host.  The caller owns every buffer; the library owns only av_ctx
 *     (constant tables, one side stream, captured graphs).  No allocation
 *     happens on the per-frame path.
 *   - `stream` is a hipStream_t passed as void*.  Calls are asynchronous and
 *     stream-ordered.  An av_ctx is bound to one device and is not thread-safe.
 *   - Batched layout: S independent video streams x W consecutive frames per
 *     call ("window").  Sequential state (track table, Kalman state, detector
 *     frame counter) lives in caller-owned device buffers and is advanced
 *     in place; S=1, W=1 is the reference's per-frame call.
 *   - Arithmetic that feeds a comparison is done in the reference's type and
 *     operation order (int32 boxes, float64 everywhere else, no FMA contraction).
 */
#ifndef AVHOT_H
#define AVHOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AV_VERSION 102

enum {
    AV_OK = 0,
    AV_EINVAL = -1,     /* bad argument (null pointer, capacity out of range, ...) */
    AV_EHIP = -2,       /* a HIP runtime call failed; see av_last_error_string()     */
    AV_ENODEV = -3,     /* no usable gfx950 device                                    */
    AV_ESTATE = -4,     /* call order violated (e.g. planner used before configure)   */
    AV_ENOMEM = -5
};

typedef struct av_ctx av_ctx;
typedef void* av_stream_t;          /* hipStream_t */

/* ---- context --------------------------------------------------------------------------------- */
int av_version(void);
const char* av_last_error_string(void);
int av_device_count(int* count);                      /* host out */
int av_ctx_create(int device, av_ctx** out);          /* fails with AV_ENODEV when no GPU is present */
int av_ctx_destroy(av_ctx* ctx);
int av_ctx_device(const av_ctx* ctx, int* device);

/* Side stream for fork/join of independent stages (tracker || Kalman->planner), also valid inside
 * a stream capture.  av_fork makes the side stream wait for everything enqueued on `main` so far;
 * av_join makes `main` wait for the side stream. */
int av_side_stream(av_ctx* ctx, av_stream_t* side);
int av_fork(av_ctx* ctx, av_stream_t main);
int av_join(av_ctx* ctx, av_stream_t main);

/* hipGraph capture of a sequence of av_* calls issued on `stream` (must not be the null stream). */
int av_graph_begin(av_ctx* ctx, av_stream_t stream);
int av_graph_end(av_ctx* ctx, av_stream_t stream, int* graph_id);
int av_graph_launch(av_ctx* ctx, int graph_id, av_stream_t stream);
int av_graph_destroy(av_ctx* ctx, int graph_id);

/* HIP events on arbitrary streams (bench.py times kernels on the stream they run on). */
int av_event_create(void** ev);
int av_event_destroy(void* ev);
int av_event_record(void* ev, av_stream_t stream);
int av_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int av_stream_sync(av_stream_t stream);
int av_stream_sync_spin(av_stream_t stream);                   /* the same, polling (no interrupt wake-up latency) */

/* Pinned host staging buffers and stream-ordered copies for the per-frame class surfaces (one packed upload and
 * one packed download per detect() / update() / step() / plan() call; demo.py:107-120 calls them once per frame).
 * av_copy_d2h with sync != 0 also waits for the stream, i.e. for the results. */
int av_host_alloc(void** p, size_t bytes);
int av_host_free(void* p);
int av_copy_h2d(void* dst_dev, const void* src_host, size_t bytes, av_stream_t stream);
int av_copy_d2h(void* dst_host, const void* src_dev, size_t bytes, av_stream_t stream, int sync);

/* ---- D1: simulated detector -------------------------------------------------------------------
 * Replaces ObjectDetector.detect -> _detect_simulated (src/perception/detector.py:86-101,125-169).
 * Detections are a pure function of (frame_count, h, w): the reference reseeds NumPy's legacy
 * MT19937 with frame_count % 1000 per frame (:134).  The kernel re-derives that stream on device.
 * frame_count[s] is the detector's counter BEFORE the window; frames frame_count[s]+1 .. +W are
 * generated and frame_count[s] += W (detector.py:96).
 *   det_n   [S][W]            number of detections (3..7)
 *   det_box [S][W][dcap][4]   x1,y1,x2,y2 (int32)
 *   det_cls [S][W][dcap]      class id 0..7
 *   det_conf[S][W][dcap]      float64
 * dcap >= 7.  status[s] != 0 if the MT19937 draw budget (227 words) was exceeded (never observed). */
int av_simdet_generate(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, int h, int w,
                       int dcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box,
                       int32_t* det_cls, double* det_conf, int32_t* status);

/* ---- K1-K3: IoU tracker -----------------------------------------------------------------------
 * Replaces MultiObjectTracker.update/_associate/_compute_iou
 * (src/tracking/multi_object_tracker.py:84-105,113-164,166-241). */
typedef struct {
    double iou_threshold;        /* 0.3  (:62) ; must be >= 0 */
    int32_t max_age;             /* 30   (:63) */
    int32_t min_hits;            /* 3    (:64) */
    int32_t trajectory_length;   /* 50   (:65) ; capacity of the per-track history ring */
} av_tracker_cfg;

/* One row of a track table, in dict-insertion (= ascending id) order.  64 bytes. */
typedef struct {
    int32_t id;
    int32_t x1, y1, x2, y2;      /* last matched bbox (:192) */
    int32_t cls;                 /* class at birth; never updated on match (:216-221) */
    int32_t age, hits, misses;
    int32_t slot;                /* index of this track's history ring in the stream's pool */
    int32_t hist_len;            /* centres appended so far (birth included); ring index = k % L */
    int32_t flags;               /* bit0: confirmed (hits >= min_hits) */
    double conf;
    float vx, vy;            /* last centre velocity (the Track.velocity property), valid when hist_len >= 2;
                                differences of half-integers, so exact in float32 */
} av_track_row;

/* Persistent per-stream tracker state (device).  Layout, for stream s with capacity tcap, L =
 * trajectory_length, all 16-byte aligned:
 *   int32 hdr[16]                  hdr[0]=n_tracks hdr[1]=next_id hdr[2]=frame_count hdr[3]=status
 *   av_track_row rows[tcap]
 *   double hist[tcap][L][4]        ring per slot: (cx, cy, vx, vy); entry k at k % L
 * trajectory  = entries max(0,hist_len-L) .. hist_len-1        (cx, cy)
 * velocities  = entries max(1,hist_len-L) .. hist_len-1        (vx, vy)
 * status bit0: table overflow (more than tcap live tracks; births were dropped, parity lost). */
size_t av_tracker_state_bytes(int tcap, int trajectory_length);
int av_tracker_reset(av_ctx* ctx, av_stream_t stream, int n_streams, int tcap, int trajectory_length,
                     void* state);
/* tcap in {64,128,...,1024}; dcap <= 64.
 *   snap    [S][W][tcap]   table after each frame (rows >= snap_n are unspecified); may be NULL
 *   snap_n  [S][W]         live rows after each frame; may be NULL iff snap is NULL
 *   det2trk [S][W][dcap]   id of the track detection j was matched to or born as (-1 beyond det_n) */
int av_tracker_update(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* cfg, int n_streams,
                      int n_frames, int dcap, const int32_t* det_n, const int32_t* det_box,
                      const int32_t* det_cls, const double* det_conf, int tcap, void* state,
                      av_track_row* snap, int32_t* snap_n, int32_t* det2trk);

/* Wire format of a track table for the all-gather of per-frame track tables across ranks (BASELINE config 5,
 * SURVEY.md section 8e; the reference has no multi-process code, F9).  What travels is what a consumer of
 * MultiObjectTracker.update()'s return value reads (multi_object_tracker.py:236-241): a 16-byte header and tcap
 * 32-byte rows, rows >= n_rows zero-filled.  Lossy against av_track_row only in `conf` (float32), `misses`
 * (saturates at 65535) and coordinates (int16: frames up to 32767 px); slot / hist_len stay local. */
#define AV_WIRE_HDR_BYTES 16
#define AV_WIRE_ROW_BYTES 32
typedef struct {
    int32_t n_rows, stream, frame, reserved;   /* global stream id; frame: see av_pack_tracks */
} av_wire_hdr;
typedef struct {
    int32_t id;
    int16_t x1, y1, x2, y2;
    int32_t age, hits;
    uint16_t misses;
    uint8_t cls, flags;          /* flags bit0: confirmed */
    float conf;
    int16_t vx2, vy2;            /* 2 x the last centre velocity (half-integers, so exact) */
} av_wire_row;
size_t av_wire_table_bytes(int tcap);          /* AV_WIRE_HDR_BYTES + tcap * AV_WIRE_ROW_BYTES */
/* Packs the tables of frames [frame_lo, frame_lo + n_sel) of every stream of a window into
 *   wire [n_streams][n_sel][av_wire_table_bytes(tcap)]
 * (n_sel = 1, frame_lo = n_frames - 1: the end-of-window table; n_sel = n_frames: every frame's table).
 * header.stream = stream0 + s (this rank's first global stream id).  header.frame = frame0 + the stream's DETECTOR FRAME COUNT
 * at that frame (ObjectDetector.frame_count after detect(), detector.py:96; 1 for a stream's first frame, plus whatever offset
 * the counter was reset to) when `frame_count` [n_streams] -- the detector's counters after the window's last frame, as
 * av_simdet_generate / av_hot_step leave them -- is given: the same value av_hot_step stamps.  frame_count NULL: frame0 + the
 * frame's index within the window.  The gather itself is torch.distributed's all_gather_into_tensor on `wire`
 * (RCCL over xGMI; distributed.TrackTableExchange). */
int av_pack_tracks(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, int tcap, int frame_lo, int n_sel,
                   int stream0, int frame0, const av_track_row* snap, const int32_t* snap_n, const int32_t* frame_count, void* wire);

/* The all-gather itself as a library call on an RCCL communicator (av_allgather_tracks(ctx, ncclComm_t, ...);
 * the reference has no counterpart).  `comm` is an ncclComm_t: the caller's own, or one made by av_comm_create from a
 * 128-byte ncclUniqueId that rank 0 obtained with av_comm_unique_id and handed to the other ranks by any means (the Python class
 * broadcasts it with torch.distributed).  librccl is opened with dlopen at the first of these calls: the library has no link-time
 * dependency on it, and a process that never gathers never loads it.
 *   send [bytes_per_rank] this rank's packed tables; recv [world * bytes_per_rank] every rank's, in rank order; stream-ordered. */
int av_comm_unique_id(void* id128);
int av_comm_create(av_ctx* ctx, const void* id128, int rank, int world, void** comm);
int av_comm_destroy(void* comm);
int av_allgather_tracks(av_ctx* ctx, void* comm, av_stream_t stream, const void* send, void* recv, size_t bytes_per_rank);

/* ---- E1-E3: vehicle state estimator ------------------------------------------------------------
 * Replaces VehicleStateEstimator.predict/update/step/_extract_state
 * (src/state_estimation/vehicle_state.py:68-198) incl. filterpy's predict/update equations. */
typedef struct {
    double dt;                   /* 0.033 (:49) */
    double process_noise;        /* 0.1   (:50) */
    double measurement_noise;    /* 1.0   (:51) */
} av_kf_cfg;

#define AV_KF_STATE_DOUBLES 48   /* x[6], P[36] row-major, prev_heading, prev_speed, time, 3 spare */
#define AV_VSTATE_DOUBLES 12     /* x y vx vy heading speed acceleration yaw_rate timestamp
                                    pos_uncertainty vel_uncertainty heading_uncertainty(=0) */
int av_kf_reset(av_ctx* ctx, av_stream_t stream, int n_streams, double* kf_state);
/* mode per (stream, frame): 0 = predict only (VehicleStateEstimator.predict)
 *                           1 = step(z)      (predict + update)
 *                           2 = step(None)   (predict + second extract)
 *                           3 = update(z) only
 * mode == NULL means 1 everywhere.
 *   z          [S][W][4]
 *   out_state  [S][W][12]
 *   plan_state [S][W][4]   (x, y, heading, speed) = the tuple demo.py:118-119 builds; may be NULL */
int av_kf_step(av_ctx* ctx, av_stream_t stream, const av_kf_cfg* cfg, int n_streams, int n_frames,
               const double* z, const uint8_t* mode, double* kf_state, double* out_state,
               double* plan_state);

/* ---- P1-P3: motion planner ---------------------------------------------------------------------
 * Replaces MotionPlanner.generate_polynomial_trajectory / evaluate_trajectory_cost / plan
 * (src/planning/motion_planner.py:126-204,206-262,264-303). */
typedef struct {
    double planning_horizon;     /* 5.0 (:69) */
    double dt;                   /* 0.1 (:70) */
    int32_t num_samples;         /* 7   (:71) lateral samples; candidates = 3 * num_samples */
    int32_t reserved;
    double w_lateral, w_velocity, w_acceleration, w_curvature;   /* 1.0 0.5 0.3 0.4 (:85-89) */
} av_planner_cfg;

#define AV_WP_DOUBLES 6          /* x y heading velocity timestamp curvature */
/* Builds the per-configuration constant tables (timestamps, 1-exp(-t), quintic blend, lateral
 * offsets) on the host and uploads them.  n_points = int(H/dt)+1 <= 1024, candidates <= 192. */
int av_planner_configure(av_ctx* ctx, const av_planner_cfg* cfg);
int av_planner_dims(const av_ctx* ctx, int* n_points, int* n_candidates);   /* host out */
/* One plan() per start state.  n_states = S*W.
 *   state     [n_states][4]          x, y, heading, speed
 *   ref_path  [n_ref][2] or NULL     shared by all states of this call (set_reference_path, :93-124)
 *   obstacles [n_obs][3] or NULL     (x, y, radius), shared by all states of this call
 *   waypoints [n_states][C][n][6]    GENERATION order (lateral outer, speed inner); may be NULL
 *   cost      [n_states][C]          generation order
 *   order     [n_states][C]          order[r] = generation index of the r-th cheapest (stable sort) */
int av_planner_plan(av_ctx* ctx, av_stream_t stream, int n_states, const double* state,
                    const double* ref_path, int n_ref, const double* obstacles, int n_obs,
                    double* waypoints, double* cost, int32_t* order);

/* generate_polynomial_trajectory for arbitrary (lateral offset, target speed) pairs (:126-204).
 *   state [n_traj][4], end_lateral_offset [n_traj], target_velocity [n_traj] -> waypoints [n_traj][n][6] */
int av_planner_generate(av_ctx* ctx, av_stream_t stream, int n_traj, const double* state,
                        const double* end_lateral_offset, const double* target_velocity,
                        double* waypoints);
/* evaluate_trajectory_cost for caller-supplied trajectories of any length (:206-262), accumulated
 * strictly left to right like the reference.  waypoints [n_traj][n_wp][6]; n_wp == 0 -> +inf (:218). */
int av_planner_evaluate(av_ctx* ctx, av_stream_t stream, int n_traj, int n_wp, const double* waypoints,
                        const double* ref_path, int n_ref, const double* obstacles, int n_obs,
                        double* cost);

/* ---- L1-L7: lane detector ----------------------------------------------------------------------
 * Replaces LaneDetector.detect and its private stages (src/perception/lane_detector.py:47-218):
 * cvtColor(BGR2GRAY) + GaussianBlur 5x5 (:66-74), median-adaptive Canny (:76-84), trapezoid ROI
 * (:47-64,86-90), HoughLinesP(1, pi/180, 50, minLineLength=50, maxLineGap=150) (:92-103), slope split
 * (:105-134), polyfit + EMA + 50-point resampling (:136-176). */
typedef struct {
    int32_t hough_threshold;     /* 50  (:98)  */
    int32_t min_line_length;     /* 50  (:99)  */
    int32_t max_line_gap;        /* 150 (:100) */
    int32_t max_segments;        /* capacity of the per-frame segment list */
    double smoothing_factor;     /* 0.7 (:45)  */
} av_lane_cfg;

/* Scratch the caller allocates once (blurred image, NMS map, union-find labels, edge maps, point list,
 * Hough accumulator, segments).  av_lane_workspace_init zero-fills it (required before first use). */
size_t av_lane_workspace_bytes(int n_streams, int h, int w, int max_segments);
int av_lane_workspace_init(av_ctx* ctx, av_stream_t stream, int n_streams, int h, int w, int max_segments,
                           void* workspace);
/* Byte range of an intermediate inside the workspace (tests, visualisation):
 * 0 blurred u8[S][h][w], 1 NMS map, 2 Canny edges (only written when stages&1), 3 ROI-masked edges
 * (consumed by the Hough stage: lines it finds are erased), 4 thresholds double[S][4] (lo, hi, median),
 * 5 segments int32[S][max_segments][4], 6 segment counts int32[S], 7 Hough accumulator. */
int av_lane_workspace_view(int what, int n_streams, int h, int w, int max_segments, size_t* offset,
                           size_t* bytes);
/*   bgr        u8 [S][h][w][3]
 *   roi_rows   int32 [h][2] inclusive column range kept per row, or NULL for the default trapezoid
 *   lane_state double [S][8]   previous smoothed fit per side: c2 c1 c0 has_prev (prev_left_fit/prev_right_fit)
 *   poly       double [S][2][3]   x = c2*y^2 + c1*y + c0   (side 0 = left, 1 = right)
 *   pts        int32 [S][2][50][2]
 *   info       int32 [S][8]    valid_left valid_right n_left_segments n_right_segments n_segments n_points lo hi
 *   conf       double [S][2]   min(1, n_side_segments / 10)
 *   stages     bit0: also write the pre-ROI Canny edge map (view 2); bit1: stop after the pixel stages
 *              (no Hough, no fit); bit4: skip the pixel stages and run Hough + fit on what the last bit1 call left
 *              in the workspace -- the two halves of a frame can then be enqueued apart, e.g. the Hough half beside
 *              the next frame's LDS-free kernels (it holds most of a CU's LDS); bit5 (with bit4): fit only, on the
 *              segment list already in the workspace (views 5 / 6) -- a test hook for the least-squares stage */
int av_lane_detect(av_ctx* ctx, av_stream_t stream, const av_lane_cfg* cfg, int n_streams, int h, int w,
                   const uint8_t* bgr, const int32_t* roi_rows, void* workspace, double* lane_state,
                   double* poly, int32_t* pts, int32_t* info, double* conf, int stages);

/* ---- D2: YOLO-mode detector ---------------------------------------------------------------------
 * Replaces ObjectDetector._detect_yolo (src/perception/detector.py:103-123), i.e. ultralytics
 * YOLO(model)(frame): letterbox -> YOLOv8n graph (Conv/BN/SiLU, C2f, SPPF, Detect+DFL) -> best-class
 * confidence filter -> class-aware NMS -> boxes scaled back to the frame.  `weights` is a HOST array in
 * the parameter order documented in perception/yolo.py (per conv: weight[cout][cin][k][k], then BN
 * gamma, beta, running_mean, running_var or, for the two plain Conv2d of each head branch, bias);
 * BatchNorm is folded and everything is converted to IEEE half at creation (float32 accumulation).  All activation and NMS scratch
 * is allocated here, once. */
typedef struct av_yolo av_yolo;
size_t av_yolo_param_count(void);
int av_yolo_create(av_ctx* ctx, int batch, int in_h, int in_w, const float* weights, size_t n_weights,
                   av_yolo** out);
/* The same with a choice of arithmetic.  AV_YOLO_FP16 (av_yolo_create): IEEE-half tensors, weights and MFMA operands, float32
 * accumulation -- the fused production path.  AV_YOLO_FP32: the reference's own precision (ultralytics runs torch float32,
 * detector.py:103-123): float32 tensors, weights and MFMA operands (v_mfma_f32_16x16x4_f32), one generic convolution kernel, no
 * fusion, no deferred tail -- the checking mode for the half-precision path and a second bench figure.  Same entry points otherwise;
 * av_yolo_tensor then returns float32 data. */
#define AV_YOLO_FP16 0
#define AV_YOLO_FP32 1
int av_yolo_create_ex(av_ctx* ctx, int batch, int in_h, int in_w, const float* weights, size_t n_weights, int precision,
                      av_yolo** out);
int av_yolo_destroy(av_yolo* h);
int av_yolo_dims(const av_yolo* h, int* net_h, int* net_w, int* n_anchors);        /* host out */
/*   bgr      u8 [batch][in_h][in_w][3] (device)
 *   det_n    int32 [batch]; det_box float [batch][max_det][4] xyxy in frame pixels (not truncated);
 *   det_conf float [batch][max_det]; det_cls int32 [batch][max_det]; rows in descending confidence */
int av_yolo_forward(av_yolo* h, av_stream_t stream, const uint8_t* bgr, float conf_thres, float iou_thres,
                    int max_det, int32_t* det_n, float* det_box, float* det_conf, int32_t* det_cls);
/* Throughput mode: with the tail deferred, av_yolo_forward enqueues decode + sort + NMS on a stream of its own behind the
 * head convolutions, so that they run beside the NEXT forward's convolutions (one workgroup per image for ~0.25 ms is the
 * latency-bound end of the chain).  Same kernels and results; det_* are complete once `stream` has passed
 * av_yolo_join_tail().  Forwards of one handle still execute in call order. */
int av_yolo_defer_tail(av_yolo* h, int enable);
int av_yolo_join_tail(av_yolo* h, av_stream_t stream);
/* The Detect head's decode (DFL expectation, best class, sigmoid; reference: ultralytics' Detect via detector.py:103-123)
 * runs in the epilogue of the head's last convolutions: the float32 logits are normally neither written nor read.  With
 * keep_logits the head writes them as well and the stand-alone decode kernel runs on them into buffers of its own
 * (tensor ids 120-122), which is how the tests show that both decodes give the same candidates bit for bit. */
int av_yolo_keep_logits(av_yolo* h, int enable);
/* Test hook: device pointer + geometry of an intermediate NHWC tensor (IEEE half; ids = yolov8.yaml layer
 * numbers, 0 = network input: RGB in 4-channel pixels inside a one-pixel frame of zeros, [H+2][W+2]; 100+2i / 101+2i =
 * float32 box / class logits of level i, valid after av_yolo_keep_logits(h, 1); 110 / 111 / 112 = the candidates of every
 * anchor as the head wrote them: box float32 [A][4], confidence float32 [A], class int32 [A]; 120-122 = the same from the
 * stand-alone decode of the kept logits). */
int av_yolo_tensor(const av_yolo* h, int id, void** ptr, int* H, int* W, int* C, int* cstride, int* coff);

/* ---- T1: maneuver tags ------------------------------------------------------
 * Replaces ManeuverDetector.detect (src/tagging/maneuver_detector.py:105-262): lateral / longitudinal /
 * turning maneuver of every frame from the current and the 14 previous vehicle states (the reference looks
 * at the newest 10 yaw rates and 15 headings of its 30-deep deques).  Enum fields are indices into the
 * reference's Enum definition order (:18-41).  `state` carries the 14 states before the window per stream. */
#define AV_MANEUVER_STATE_DOUBLES 32   /* frames seen, 14 yaw rates, 14 headings (oldest first), 3 spare */
typedef struct av_maneuver_row {
    int32_t lateral, longitudinal, turning, reserved;
    double lateral_confidence, longitudinal_confidence, turning_confidence;
    double speed_kmh, acceleration, yaw_rate_deg, timestamp;
} av_maneuver_row;                      /* 72 bytes */
int av_maneuver_reset(av_ctx* ctx, av_stream_t stream, int n_streams, double* state);
/*   vstate      f64 [n_streams][n_frames][AV_VSTATE_DOUBLES]  (the out_state of av_kf_step)
 *   lane_offset f64 [n_streams][n_frames], NaN = none for that frame; may be NULL (:197-203)
 *   out         av_maneuver_row [n_streams][n_frames] */
int av_maneuver_detect(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, const double* vstate,
                       const double* lane_offset, double* state, av_maneuver_row* out);

/* ---- T2: interaction tags ----------------------------------------------------
 * Replaces InteractionDetector.detect (src/tagging/interaction_detector.py:132-222) and its helpers
 * (_estimate_distance :224, _estimate_relative_speed :250, _calculate_ttc :262, _analyze_interaction :270,
 * _calculate_overall_risk :374): the consumer of the per-frame track tables.  The reference is handed the
 * tracker's returned (confirmed) tracks in order; here those are the snapshot rows with flags bit0 set.  A
 * track's centre-x history (deque of 30, needed by the cut-in rule) lives in `state`, indexed by the row's
 * history slot, which is stable for a track's lifetime.  Types / risks are indices into the reference's Enum
 * definition order (:20-40); type -1 = no interaction for that row. */
typedef struct {
    int32_t frame_h, frame_w;        /* frame_shape (:135), e.g. 720, 1280 */
    int32_t class_kind[16];          /* per class id: 0 other, 1 pedestrian, 2 cyclist/bicycle, 3 car/truck/bus,
                                        4 motorcycle (counted as a vehicle, :160, but outside the vehicle rules) */
} av_interaction_cfg;
typedef struct av_interaction_row {
    int32_t type, risk, agent_id, cls;
    double confidence, distance, relative_speed, ttc;       /* ttc NaN = None */
} av_interaction_row;                /* 48 bytes */
typedef struct av_interaction_summary {
    int32_t agent_count, pedestrian_count, cyclist_count, vehicle_count;
    int32_t n_interactions, primary_type, overall_risk, primary_row;   /* primary_type -1 = None */
    double closest_distance, min_ttc, timestamp;                        /* min_ttc NaN = None */
} av_interaction_summary;            /* 56 bytes */
size_t av_interaction_state_bytes(int tcap);
int av_interaction_reset(av_ctx* ctx, av_stream_t stream, int n_streams, int tcap, void* state);
/*   snap, snap_n   the tracker's per-frame tables (av_tracker_update), tcap == 64.  Windows longer than 64 frames
 *                  are evaluated as independent 64-frame chunks that rebuild the 30-deep histories from the 29
 *                  frames before them; this relies on what the tracker guarantees -- a returned track is returned
 *                  in every frame until it is deleted.  Tables without that property: windows of <= 64 frames.
 *   vstate         f64 [S][W][AV_VSTATE_DOUBLES] (ego speed = element 5); NULL = 10.0 (:166)
 *   has_state      u8 [S][W], 0 = vehicle_state was None for that frame; NULL = always given
 *   vy             f64 [S][W][tcap] image-space y velocity per row instead of the row's float vy; may be NULL
 *   rows           av_interaction_row [S][W][tcap], aligned with the snapshot rows
 *   summary        av_interaction_summary [S][W] */
int av_interaction_detect(av_ctx* ctx, av_stream_t stream, const av_interaction_cfg* cfg, int n_streams, int n_frames,
                          int tcap, const av_track_row* snap, const int32_t* snap_n, const double* vstate,
                          const uint8_t* has_state, const double* vy, void* state, av_interaction_row* rows,
                          av_interaction_summary* summary);

/* ---- rendering ----------------------------------------------------------------
 * Replaces the cv2 drawing behind BEVRenderer.render (src/visualization/bev_renderer.py:286-348 and its helpers
 * :92-284,350-364), OverlayRenderer (src/visualization/overlays.py:26-210) and the draw_* methods of the four hot-path
 * classes (detector.py:171, lane_detector.py:220, multi_object_tracker.py:251, motion_planner.py:305).  A picture is an
 * ORDERED list of primitives (later ones paint over earlier ones), rasterised per pixel by exact integer tests -- the
 * geometry of the reference's calls, not OpenCV's scan conversion, labels in a 5x7 bitmap font: parity unpinned
 * (OpenCV absent), bit-exact against oracle/raster_ref.py. */
enum {
    AV_PRIM_RECT = 1,        /* filled, corners (x0,y0)-(x1,y1) inclusive                                   */
    AV_PRIM_SEG = 2,         /* segment (x0,y0)-(x1,y1), thickness p: pixels within p/2 of it               */
    AV_PRIM_QUAD = 3,        /* filled convex quadrilateral (x0,y0) (x1,y1) (x2,y2) (x3,y3), edges included */
    AV_PRIM_DISC = 4,        /* filled circle, centre (x0,y0), radius p                                     */
    AV_PRIM_RING = 5,        /* circle outline of thickness 1, centre (x0,y0), radius p                     */
    AV_PRIM_GLYPH = 6,       /* character p (ASCII 32..126) of the 5x7 font, top-left (x0,y0), scale x1     */
    AV_PRIM_BLEND_RECT = 7,  /* rectangle blended 0.7 picture + 0.3 colour (cv2.addWeighted panels)         */
    AV_PRIM_POLY_BLEND = 8   /* polygon verts[x0 .. x0+y0), even-odd, blended the same; (x2,y2)-(x3,y3) = its bounding box */
};
typedef struct {
    int32_t type;
    int32_t x0, y0, x1, y1, x2, y2, x3, y3;
    int32_t p;
    uint8_t b, g, r, a;      /* colour in OpenCV's channel order; a unused */
    int32_t reserved;
} av_prim;                   /* 48 bytes */
/*   img     u8 [n_images][h][w][3], drawn in place
 *   prims   av_prim [n_images][prim_cap], n_prims int32 [n_images]
 *   verts   int32 [n_images][vert_cap][2] polygon vertices (x, y); may be NULL when no POLY primitive is used */
int av_raster_draw(av_ctx* ctx, av_stream_t stream, int n_images, int h, int w, uint8_t* img, const av_prim* prims, int prim_cap,
                   const int32_t* n_prims, const int32_t* verts, int vert_cap);
/* The BEV panel of BEVRenderer.render (bev_renderer.py:286-348) as a primitive list built ON THE DEVICE from the hot
 * loop's tables -- planner waypoints + stable order, tracker snapshot rows + history rings, Kalman output -- for frame
 * `frame` of the window of every stream; paint it with av_raster_draw over the road image.  Layering as the reference:
 * candidates of rank 1 .. n_candidates-1 (grey), the planned path (rank 0, green, a disc on every third waypoint), the
 * confirmed tracks (footprint, outline, heading arrow, "ID:n", trail), the ego vehicle + uncertainty circle, the legend.
 * The list has a fixed layout of av_bev_prim_cap() slots per stream; unused slots are type 0. */
typedef struct {
    int32_t width, height;          /* 600, 600 (bev_renderer.py:29-33) */
    double pixels_per_meter;        /* 10.0 */
    double x_min, x_max, y_min, y_max;   /* (-30, 30), (-10, 50) */
    int32_t n_candidates;           /* how many of the ranked candidates to show (demo.py:143 passes candidate_trajs[:10]) */
    int32_t reserved;
} av_bev_cfg;
int av_bev_prim_cap(const av_bev_cfg* cfg, int tcap, int n_points);
/*   snap, snap_n   [S][W][tcap], [S][W] of av_tracker_update; tracker_state its persistent state (history rings)
 *   vstate         [S][W][AV_VSTATE_DOUBLES] of av_kf_step, or NULL (no ego vehicle)
 *   waypoints      [S*W][C][n][6], order [S*W][C] of av_planner_plan
 *   prims          av_prim [S][prim_cap], n_prims int32 [S]
 * The history rings hold the trails as of the END of the window: build panels for frame = W-1 (or run W = 1). */
int av_bev_build(av_ctx* ctx, av_stream_t stream, const av_bev_cfg* cfg, int n_streams, int n_frames, int frame, int tcap,
                 int trajectory_length, const av_track_row* snap, const int32_t* snap_n, const void* tracker_state,
                 const double* vstate, const double* waypoints, const int32_t* order, av_prim* prims, int prim_cap, int32_t* n_prims);
/* Bilinear resize (half-pixel centres) of src [sh][sw][3] into the dh x dw window at column dst_x0 of a picture whose
 * rows hold dst_pitch_px pixels: OverlayRenderer.create_side_by_side's cv2.resize + hstack (overlays.py:187-196). */
int av_resize_into(av_ctx* ctx, av_stream_t stream, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, int dst_pitch_px,
                   int dst_x0);

/* ---- video ingest ----------------------------------------------------------------
 * The pixel half of VideoDataLoader.read_frame / read_frame_at (data/loaders/video_loader.py:88-131): what a decoder hands over
 * as planar YUV 4:2:0 becomes the BGR frame cv2.VideoCapture.read returns, then av_resize_into scales it to target_size
 * (:103-104).  Bitstream decoding is NOT here (no decoder in this image): the loader reads uncompressed containers.
 *   yuv  u8 [n_frames][h*w*3/2]  I420: Y plane, then U, then V (each (h/2) x (w/2));   bgr  u8 [n_frames][h][w][3] */
int av_i420_to_bgr(av_ctx* ctx, av_stream_t stream, int n_frames, int h, int w, const uint8_t* yuv, uint8_t* bgr);

/* ---- synthetic input ---------------------------------------------------
 * Deterministic 8-bit BGR road scenes generated on the device, standing in for the reference's lost
 * SyntheticDataGenerator (data/generators, source absent).  Frame (stream0+s, frame) is bit-identical to
 * oracle/lane_ref.py: synthetic_frame().   bgr: u8 [n_streams][h][w][3] */
int av_synth_frames(av_ctx* ctx, av_stream_t stream, int n_streams, int h, int w, int stream0, int frame,
                    uint8_t* bgr);

/* ---- one launch per time-step (BASELINE config 4, window 1; the reference's per-frame cadence, demo.py:97-120) -----------
 * av_hot_step = av_simdet_generate + av_tracker_update + av_kf_step + av_planner_plan (no reference path / obstacles) for ONE
 * frame of every stream, as a single kernel with role-split workgroups that run the stage kernels' own device code: the
 * results are those of the four calls bit for bit.  Buffers as in the stage calls with n_frames = 1:
 *   frame_count [S]; det_n [S], det_box [S][dcap][4], det_cls / det_conf [S][dcap], det_status [S] or NULL;
 *   tracker_state as av_tracker_update; snap [S][tcap] + snap_n [S] (or both NULL); det2trk [S][dcap];
 *   z [S][4]; kf_state [S][AV_KF_STATE_DOUBLES]; vstate [S][AV_VSTATE_DOUBLES]; plan_state [S][4];
 *   waypoints [S][C][n][6] or NULL; cost, order [S][C];
 *   wire: NULL, or [S][av_wire_table_bytes(tcap)] -- every stream's table in the all-gather's wire format (av_pack_tracks with
 *   n_sel = 1), written by the same launch (needs snap); header.stream = stream0 + s, header.frame = frame0 + frame_count[s]
 *   after the step (the stream's own detector frame count, read on the device: a captured graph stamps every replay correctly).
 * Built for tcap 64, dcap 7..8, iou_threshold > 0 (AV_EINVAL otherwise: use the stage calls). */
int av_hot_step(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tracker_cfg, const av_kf_cfg* kf_cfg, int n_streams, int h,
                int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n, int32_t* det2trk,
                const double* z, double* kf_state, double* vstate, double* plan_state, double* waypoints, double* cost,
                int32_t* order, void* wire, int stream0, int frame0);

/* Consecutive time-steps OVERLAPPED (still one launch per step, same results bit for bit).  The reference's loop runs frame t + 1
 * after frame t (demo.py:97-120); what frame t + 1 needs of frame t is the stream's tracker table (tracker role) and its filter
 * state (Kalman role), not the planner's output.  The caller launches step `seq` (0, 1, 2, ... since the state was reset) on HIP
 * stream seq % D of D = 2 .. 4 streams -- so that step seq + D follows step seq in stream order -- and gives the D steps that may
 * be in flight DIFFERENT per-step buffers (det_*, snap, snap_n, det2trk, z, vstate, plan_state, waypoints, cost, order, wire); the persistent
 * buffers (frame_count, det_status, tracker_state, kf_state) and seq_flags are shared.  On the device a role of step `seq` waits
 * until seq_flags says its stream's role of step seq - 1 has finished and published its state:
 *   seq_flags  int32 [AV_STEP_FLAG_INTS(S)], set up when the state is reset: [32 (2 s + r)] = 0, the steps done by role r (0 tracker,
 *              1 Kalman) of stream s, a 128-byte line each; [64 S] = 0, the fault word -- bit 0: a workgroup's wait ran out
 *              (AVHOT_STEP_SPIN polls, default 2^22: a predecessor that was never launched) and it left without running its step,
 *              bit 1: frame_count[s] was not base + seq; check it after synchronising; [64 S + 32 + s] = frame_count[s] at the
 *              reset (the detections of step seq are made for frame count base + seq + 1 without waiting for step seq - 1);
 *              the last 64 ints: phase clocks summed by the kernel when AVHOT_STEP_FENCE=8 (tools/steptime.py), otherwise unused.
 * Step numbers are 32-bit and wrap (a signed int carrying an unsigned count: 2^31 - 1 is followed by -2^31, -1 by 0); the stream and
 * buffer set of step seq are those of (uint32_t)seq % D.
 * `depth` = D: up to D launches may be in flight, each possibly waiting for the one before it, so all of them must be RESIDENT
 * together: the call picks sixteen or eight waves per workgroup accordingly and returns AV_EINVAL when D launches of 2 S workgroups
 * cannot fit (64 streams: D <= 2 with sixteen waves, <= 4 with eight).  HotLoop(window=1, overlap=D) drives it. */
#define AV_STEP_FLAG_INTS(n_streams) (65 * (n_streams) + 32 + 64)
int av_hot_step_seq(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tracker_cfg, const av_kf_cfg* kf_cfg, int n_streams, int h,
                    int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                    double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n, int32_t* det2trk,
                    const double* z, double* kf_state, double* vstate, double* plan_state, double* waypoints, double* cost,
                    int32_t* order, void* wire, int stream0, int frame0, int32_t* seq_flags, int seq, int depth);

/* n_steps consecutive overlapped steps (seq0, seq0 + 1, ...) enqueued by ONE call: the launch loop of av_hot_step_seq in C (a Python
 * caller spends 8 us per launch, the device 6-7), with `depth` (2 .. AV_STEP_MAX_DEPTH) steps in flight: step q runs on
 * streams[q % depth] and writes sets[q % depth], so after the call the sets hold the outputs of the last `depth` steps.  All launches
 * in flight have to fit on the device together (depth * 2 S workgroups, two per CU): AV_EINVAL otherwise.
 *   z_steps     NULL (every step reads its set's z), or [n_steps][S][4]: the measurements of step seq0 + i at z_steps + i * S * 4
 *   wire_steps  NULL, or [n_steps][S][av_wire_table_bytes(tcap)]: every step's wire tables (the per-frame all-gather's payload) */
#define AV_STEP_MAX_DEPTH 4
typedef struct av_step_set {
    int32_t *det_n, *det_box, *det_cls;
    double* det_conf;
    av_track_row* snap;
    int32_t *snap_n, *det2trk;
    const double* z;
    double *vstate, *plan_state, *waypoints, *cost;
    int32_t* order;
} av_step_set;
int av_hot_steps_seq(av_ctx* ctx, int depth, const av_stream_t* streams, const av_tracker_cfg* tracker_cfg, const av_kf_cfg* kf_cfg,
                     int n_streams, int h, int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_status, void* tracker_state,
                     double* kf_state, const av_step_set* sets, const double* z_steps, void* wire_steps, int stream0, int frame0,
                     int32_t* seq_flags, int seq0, int n_steps);

#ifdef __cplusplus
}
#endif
#endif /* AVHOT_H */
