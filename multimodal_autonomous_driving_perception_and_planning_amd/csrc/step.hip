// H: one TIME-STEP of the hot loop for every stream in ONE launch (BASELINE config 4: "64 concurrent synthetic streams batched
// through detector -> tracker -> KF -> planner, hipGraph-captured per-step"; the reference's per-frame cadence, demo.py:97-120).
//
// With one frame per stream and launch the four stage kernels are latency chains of a few microseconds each, and a step is
// their launches: four graph nodes, ~60 us per step of 64 frames, of which ~16 us graph replay and 23 us a planner kernel for 64
// start states.  Here the step is one kernel with role-split workgroups, every role running the stage kernels' OWN device code
// (simdet_frame, tracker_body, kf_axis_body / kf_dense_stream, plan_block: this file includes the stage files for their device
// parts), so the results are those of the separate launches bit for bit (tests/test_gpu_step.py):
//   workgroups [0, S)      stream s: simulated detections of its next frame (one thread: a 232-byte table row + box arithmetic),
//                          then the tracker frame on eight replica waves, then -- optionally -- the stream's 32-byte-per-row
//                          wire table for the all-gather (exchange.hip's format), all from the same workgroup
//   workgroups [S, 2 S)    stream s: Kalman step in the first wave (axis-separable filter; the dense filter on one lane for a
//                          stream flagged non-separable), then the 3 C candidate trajectories spread over the workgroup's
//                          eight waves (plan_block<1, 8>), cost ranking, outputs
// The two roles of a stream never exchange data (the planner does not consume tracks).
#define AVHOT_DEVICE_ONLY
#include "simdet.hip"
#include "tracker.hip"
#include "kf.hip"
#include "planner.hip"
#undef AVHOT_DEVICE_ONLY

namespace {

struct StepArgs {
    int S, h, w, dcap, tcap;
    av_tracker_cfg tcfg;
    av_kf_cfg kcfg;
    PlanParams pp;
    int32_t* frame_count; const SimRow* tab; const double* cdf;
    int32_t *det_n, *det_box, *det_cls; double* det_conf; int32_t* det_status;
    unsigned char* trk_state; av_track_row* snap; int32_t* snap_n; int32_t* det2trk;
    const double* z; double *kf_state, *vstate, *plan_state;
    double *wp, *cost; int32_t* order;
    uint8_t* wire; int stream0, frame0;
};

constexpr int STEP_NW = 8;

// PW: waves of a planner workgroup (the tracker role always runs on STEP_NW = 8; with PW = 16 its workgroups' other eight waves leave
// at once).  The 3 C = 21 trajectories of a start state are dealt to the waves whole: eight waves take three rounds, sixteen two.
template <int PW>
__global__ void __launch_bounds__(PW * 64) hot_step_kernel(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.S) {
        if (PW > STEP_NW && tid >= STEP_NW * 64) return;
        const int s = blockIdx.x;
        if (tid == 0)
            simdet_frame<true>(s, s, 0, a.h, a.w, a.dcap, a.frame_count, a.tab, a.cdf, a.det_n, a.det_box, a.det_cls, a.det_conf, a.det_status);
        __syncthreads();              // the detections are in memory and visible to this workgroup (fence + vmcnt(0))
        tracker_body<false, 8, STEP_NW>(a.tcfg, 1, a.dcap, a.det_n, a.det_box, a.det_cls, a.det_conf, a.tcap, a.trk_state, a.snap, a.snap_n,
                                        a.det2trk, 1, s, smem);
        if (a.wire) {                 // this stream's table in wire format (pack_tracks_kernel's row conversion)
            __syncthreads();          // the snapshot rows the bookkeeper wave wrote
            // frame = frame0 + the stream's detector frame count after this step: read from memory, so that a captured graph
            // (fixed kernel arguments) stamps every replay with its own index
            if (tid < a.tcap)
                wire_put(a.wire + (size_t)s * (AV_WIRE_HDR_BYTES + (size_t)a.tcap * AV_WIRE_ROW_BYTES), tid, a.snap_n[s], a.tcap,
                         a.snap + (size_t)s * a.tcap, a.stream0 + s, a.frame0 + a.frame_count[s]);
        }
    } else {
        const int s = blockIdx.x - a.S;
        if (tid < 64) {
            const bool separable = kf_axis_body(a.kcfg, 1, a.z, nullptr, a.kf_state, a.vstate, a.plan_state, s, tid);
            if (!separable && tid == 0) kf_dense_stream_lds(a.kcfg, s, 1, a.z, nullptr, a.kf_state, a.vstate, a.plan_state);   // (LDS form: kf_dense.inc)
        }
        __syncthreads();              // the planner's start state (plan_state[s]) is in memory and visible to this workgroup
        plan_block<1, PW>(a.pp, s, a.S, a.plan_state, nullptr, 0, nullptr, 0, a.wp, a.cost, a.order, reinterpret_cast<double*>(smem));
    }
}

}  // namespace

extern "C" int av_hot_step(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg, int n_streams, int h,
                           int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                           double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n,
                           int32_t* det2trk, const double* z, double* kf_state, double* vstate, double* plan_state,
                           double* waypoints, double* cost, int32_t* order, void* wire, int stream0, int frame0) {
    AV_REQUIRE(ctx && tcfg && kcfg && frame_count && det_n && det_box && det_cls && det_conf && tracker_state && z && kf_state &&
                   vstate && plan_state && cost && order,
               AV_EINVAL, "av_hot_step: null argument");
    AV_REQUIRE(n_streams > 0, AV_EINVAL, "av_hot_step: n_streams must be > 0");
    AV_REQUIRE(ctx->planner_ready && ctx->d_simtab, AV_ESTATE, "av_hot_step: call av_planner_configure first");
    AV_REQUIRE((snap == nullptr) == (snap_n == nullptr), AV_EINVAL, "av_hot_step: snap and snap_n go together");
    AV_REQUIRE(!wire || snap, AV_EINVAL, "av_hot_step: the wire tables are made from the snapshot rows");
    // the shapes the one-launch step is built for; anything else keeps the four stage calls
    AV_REQUIRE(tcap == 64 && dcap >= 7 && dcap <= 8 && tcfg->iou_threshold > 0.0 && tcfg->trajectory_length >= 1, AV_EINVAL,
               "av_hot_step: needs tcap 64, dcap 7..8 and iou_threshold > 0 (use the stage calls otherwise)");
    AV_REQUIRE(h > 0 && w > 121, AV_EINVAL, "av_hot_step: frame %dx%d too small", w, h);
    StepArgs a{};
    a.S = n_streams, a.h = h, a.w = w, a.dcap = dcap, a.tcap = tcap;
    a.tcfg = *tcfg, a.kcfg = *kcfg;
    fill_params(ctx, a.pp);
    a.frame_count = frame_count, a.tab = (const SimRow*)ctx->d_simtab, a.cdf = ctx->d_cdf;
    a.det_n = det_n, a.det_box = det_box, a.det_cls = det_cls, a.det_conf = det_conf, a.det_status = det_status;
    a.trk_state = (unsigned char*)tracker_state, a.snap = snap, a.snap_n = snap_n, a.det2trk = det2trk;
    a.z = z, a.kf_state = kf_state, a.vstate = vstate, a.plan_state = plan_state;
    a.wp = waypoints, a.cost = cost, a.order = order;
    a.wire = (uint8_t*)wire, a.stream0 = stream0, a.frame0 = frame0;
    // dynamic LDS: the larger of the tracker's (av_tracker_update's layout for one staged frame, eight replicas) and the planner's
    const int fc = 1;
    const size_t chunk_bytes = (size_t)((fc + 3) & ~3) * 4 + (size_t)fc * dcap * 16 + (((size_t)fc * dcap + 1) & ~size_t(1)) * 4 +
                               (size_t)fc * dcap * 16 + 16;
    const size_t rep_bytes = ((sizeof(Shared) + 63) & ~size_t(63)) + (size_t)tcap * sizeof(av_track_row);
    const size_t lds_t = rep_bytes * STEP_NW + chunk_bytes + 2 * 576;
    const char* pwe = getenv("AVHOT_STEP_PW");
    const int pw = pwe && atoi(pwe) == 8 ? 8 : 16;
    const size_t lds_p = plan_lds_doubles(1, ctx->n_points, ctx->n_cand, pw) * 8;
    const size_t lds = lds_t > lds_p ? lds_t : lds_p;
    // static __shared__ of the kernel (the Kalman bodies' arrays) counts against the same 64 KB
    static const size_t lds_static = [] {
        hipFuncAttributes fa{};
        return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(hot_step_kernel<16>)) == hipSuccess ? (size_t)fa.sharedSizeBytes : (size_t)16384;
    }();
    AV_REQUIRE(lds + lds_static <= 64 * 1024, AV_EINVAL, "av_hot_step: configuration needs %zu B of dynamic + %zu B of static LDS (limit 65536)",
               lds, lds_static);
    if (pw == 16) hipLaunchKernelGGL(hot_step_kernel<16>, dim3(2 * n_streams), dim3(16 * 64), lds, as_stream(stream), a);
    else hipLaunchKernelGGL(hot_step_kernel<8>, dim3(2 * n_streams), dim3(8 * 64), lds, as_stream(stream), a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
