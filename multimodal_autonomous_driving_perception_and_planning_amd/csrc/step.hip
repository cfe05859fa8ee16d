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
//                          sixteen (or eight) waves (plan_block<1, PW>), cost ranking, outputs
// The two roles of a stream never exchange data (the planner does not consume tracks).
// av_hot_step launches one such step; av_hot_step_seq / av_hot_steps_seq keep up to four consecutive steps in flight (below).
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
    int* flags; int seq, spin, fence, stage_off;      // consecutive steps overlapped (av_hot_step_seq): see seq_enter
};

constexpr int STEP_NW = 8;

// ---- consecutive time-steps overlapped --------------------------------------------------------------------------------------------
// One launch per time-step leaves the chip to ONE kernel of 2 S workgroups whose 13 us are mostly latency (launch, first loads,
// the Kalman -> arc length -> trajectories chain, the drain of the stores), and step t + 1 only starts when step t has drained.
// What step t + 1 really needs of step t is less: its tracker role the stream's tracker table, its Kalman role the stream's
// filter state -- not the planner's 4.5 MB of waypoints.  av_hot_step_seq therefore lets the caller launch step q on HIP stream
// q mod D of D = 2..4 streams (step q + D follows step q in stream order) and orders step q + 1 behind step q per stream and role
// on the device: a counter per stream and role (0 tracker, 1 Kalman) holds the number of steps whose role has finished.  A role of
// step q waits until its counter reads q and publishes q + 1 when its persistent state is written.
// Steps land on different XCDs (measured: the predecessor's role had run on another XCD in 99.98 % of 537 600 hand-overs), each
// with an L2 of its own, so the hand-over has to go through memory:
//   * an agent-scope ACQUIRE in the consumer would be buffer_inv sc1, which drops the XCD's whole L2 -- that alone takes the
//     step from 7.9 to 15 us (every table of every workgroup is then re-read from HBM), and an agent-scope RELEASE in the publisher
//     (buffer_wbl2) waits for the XCD's dirty lines, the planner's waypoints among them.  Neither is used.
//   * Instead the few bytes that cross a step boundary -- tracker: header + rows (4 160 B) and the frame counter; Kalman: 46 doubles
//     -- are WRITTEN with device-scope stores (sc1: written through to memory) and READ once, into LDS, with device-scope loads (sc1:
//     never served from the CU's L1 or from an XCD's possibly stale L2 line); the stage code runs on the LDS copy.  Nothing else a
//     role reads was written by the previous step.  The counter is stored by the wave that wrote the record (the tracker's row-keeping
//     wave, the Kalman wave) behind an explicit s_waitcnt vmcnt(0): its stores are acknowledged, i.e. visible device-wide
//     (__syncthreads() is s_waitcnt lgkmcnt(0) + s_barrier on this target and waits for no global store); the consumer's loads are
//     issued after its poll has returned the new count.  tests/test_gpu_step.py: 150 unsynchronised steps x 64 streams at depth 2, 3, 4
//     bit-identical to the serial loop; tools/soak.py: 10^6 steps.
// The per-step outputs (detections, snapshot rows, det2trk, Kalman output, waypoints, costs, order, wire table) rotate through D
// buffer sets on the host side, so steps in flight never write the same output and the Kalman counter moves on before the
// planner has run.  All launches in flight must be RESIDENT together (each may be waiting for the one before it): hot_step_args picks
// sixteen or eight waves per workgroup from the occupancy and refuses a depth that does not fit.  Every wait is bounded: after `spin`
// polls the workgroup sets the fault word (bit 0) and leaves without running its step, the launches behind it give up at once
// (HotLoop.synchronize raises) -- no launch can hang on a lost predecessor.
// seq_flags layout (AV_STEP_FLAG_INTS): one 128-byte line per counter -- 2 S pollers hammer them -- then the fault word's line, then
// the streams' detector frame counts at reset (the count before step q is base + q: the detections do not have to wait)
__device__ __host__ inline int flag_fault(int S) { return 64 * S; }
__device__ __host__ inline int flag_base(int S) { return 64 * S + 32; }
__device__ __host__ inline int flag_stats(int S) { return 65 * S + 32; }      // 32 u64 of phase clocks (AVHOT_STEP_FENCE=8, tools/steptime.py)
struct StepClock {                    // debug only: 100-MHz clock stamps of a role's thread 0, summed per phase over all launches
    unsigned long long t;
    bool on;
    unsigned long long* acc;
    __device__ void start(const StepArgs& a, int role);
    __device__ void mark(int k) {
        if (!on) return;
        const unsigned long long n = __builtin_amdgcn_s_memrealtime();
        atomicAdd(acc + k, n - t);
        t = n;
    }
};

__device__ void StepClock::start(const StepArgs& a, int role) {
    on = a.flags && (a.fence & 8) && threadIdx.x == 0;
    if (!on) return;
    acc = reinterpret_cast<unsigned long long*>(a.flags + flag_stats(a.S)) + role * 16;
    t = __builtin_amdgcn_s_memrealtime();
    atomicAdd(acc + 15, 1ull);
}

__device__ __forceinline__ bool seq_enter(const StepArgs& a, int slot, int* go) {
    if (threadIdx.x == 0) {
        int ok = 0;
        // (locals: with the arguments read through `a` inside the loop the compiler reloads them from the kernel-argument segment on
        // every poll, a scalar-cache round trip in front of each device-scope load)
        int* const counter = a.flags + 32 * slot;
        int* const fault = a.flags + flag_fault(a.S);
        const int want = a.seq, spin = a.spin;
        for (int n = 0; n <= spin; ++n) {
            // (step numbers are 32-bit and wrap -- three hours of stepping at this rate -- so "has reached" is a signed distance)
            if ((int)((unsigned)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)want) >= 0) { ok = 1; break; }
            // (once any wait has run out the chain is broken for good: the launches behind it give up at once instead of one timeout each)
            if ((n & 255) == 255 && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
        if (!ok) atomicOr(a.flags + flag_fault(a.S), 1);
        if (ok && (a.fence & 8) && a.seq != 0) {      // debug: publisher's clock at its counter store -> this poll's return
            const unsigned long long tp = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.flags + 32 * slot + 2), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            unsigned long long* acc = reinterpret_cast<unsigned long long*>(a.flags + flag_stats(a.S)) + (slot & 1) * 16;
            if (now > tp) atomicAdd(acc + 8, now - tp), atomicAdd(acc + 9, 1ull);
        }
        *go = ok;
    }
    __syncthreads();                  // (also keeps the compiler from moving any load of the role above the poll)
    if (!*go) return false;
    if (a.fence & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // debug: the form the comment above prices
    return true;
}
// by ONE thread, behind a workgroup barrier that follows the role's last (device-scope) store to its persistent state
__device__ __forceinline__ void seq_leave(const StepArgs& a, int slot) {
    if (a.fence & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (a.fence & 8)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.flags + 32 * slot + 2), __builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.flags + 32 * slot, (int)((unsigned)a.seq + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// n8 8-byte words of a stream's record into LDS, by the first `nthreads` threads of the workgroup.  coherent: device-scope loads (the
// record was written by the previous step, possibly on another XCD).
__device__ __forceinline__ void fetch_record(const void* src, void* dst_lds, int n8, int first, int end, bool coherent) {
    unsigned long long* g = const_cast<unsigned long long*>(reinterpret_cast<const unsigned long long*>(src));
    unsigned long long* l = reinterpret_cast<unsigned long long*>(dst_lds);
    if ((int)threadIdx.x < first || (int)threadIdx.x >= end) return;
    for (int i = (int)threadIdx.x - first; i < n8; i += end - first)
        l[i] = coherent ? __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g[i];
}

// PW: waves of a planner workgroup (the tracker role always runs on STEP_NW = 8; with PW = 16 its workgroups' other eight waves leave
// at once).  The 3 C = 21 trajectories of a start state are dealt to the waves whole: eight waves take three rounds, sixteen two.
template <int PW>
__global__ void __launch_bounds__(PW * 64) hot_step_kernel(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int go, fc_stage;
    __shared__ __attribute__((aligned(16))) double kf_stage[AV_KF_STATE_DOUBLES + 2];
    // the step's inputs that do not come from the previous step, in LDS before the wait: the frame's detections (made here, copied
    // to their output arrays by the other threads) and the ego measurement
    __shared__ __attribute__((aligned(16))) int d_box[8 * 4];
    __shared__ __attribute__((aligned(16))) double d_conf[8], d_area[8], z_stage[4];
    __shared__ int d_cls[8], d_n[1];
    const int tid = threadIdx.x;
    const bool seq = a.flags != nullptr;          // consecutive steps overlapped: wait for / publish to the neighbouring launches
    // Both roles run on an LDS copy of the stream's record (tracker: header + rows; Kalman: the filter's 46 doubles), fetched by the
    // whole workgroup at once -- with device-scope loads and stores when the neighbouring steps are separate launches in flight.
    if ((int)blockIdx.x < a.S) {
        if (PW > STEP_NW && tid >= STEP_NW * 64) return;
        const int s = blockIdx.x;
        StepClock ck;
        ck.start(a, 0);
        // the detections first: overlapped, the detector's count before step q is its count at reset + q -- no need to wait for step
        // q - 1 (tid 0 checks that against the counter the predecessor left: fault bit 1)
        int fc_before = 0;
        if (tid == 0) {
            fc_before = seq ? (int)((unsigned)a.flags[flag_base(a.S) + s] + (unsigned)a.seq) : a.frame_count[s];
            fc_stage = fc_before;
            simdet_frame<true>(0, 0, 0, a.h, a.w, a.dcap, &fc_stage, a.tab, a.cdf, d_n, d_box, d_cls, d_conf,
                               a.det_status ? a.det_status + s : nullptr);
            for (int i = 0; i < a.dcap; ++i)      // the boxes' areas, as the tracker's chunk hand-over makes them (exact in float64)
                d_area[i] = (double)(d_box[4 * i + 2] - d_box[4 * i]) * (double)(d_box[4 * i + 3] - d_box[4 * i + 1]);
        }
        ck.mark(0);                   // detections made
        if (seq && !seq_enter(a, 2 * s, &go)) return;
        ck.mark(1);                   // waited for the predecessor
        unsigned char* stage = smem + a.stage_off;
        int fc0 = 0;
        if (seq && tid == 0) fc0 = __hip_atomic_load(a.frame_count + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (in flight with the table)
        // (by the waves thread 0 is not in: launched serially, the record comes in while thread 0 still makes the detections)
        fetch_record(a.trk_state + (size_t)s * state_bytes(a.tcap, a.tcfg.trajectory_length), stage,
                     (HDR_INTS * 4 + a.tcap * (int)sizeof(av_track_row)) / 8, 64, STEP_NW * 64, seq);
        if (tid == 0) {
            if (seq) {
                if (fc0 != fc_before) atomicOr(a.flags + flag_fault(a.S), 2);
                __hip_atomic_store(a.frame_count + s, fc_stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                a.frame_count[s] = fc_stage;
            }
        }
        if (seq) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): thread 0's frame-counter store is acknowledged before any wave passes the barrier
        __syncthreads();              // the detections and the table copy are in LDS
        ck.mark(2);                   // record in LDS
        if (tid < a.dcap) {           // the detections' output arrays (the tracker reads the LDS copy)
            reinterpret_cast<int4*>(a.det_box)[(size_t)s * a.dcap + tid] = reinterpret_cast<const int4*>(d_box)[tid];
            a.det_cls[(size_t)s * a.dcap + tid] = d_cls[tid];
            a.det_conf[(size_t)s * a.dcap + tid] = d_conf[tid];
            if (tid == 0) a.det_n[s] = d_n[0];
        }
        tracker_body<false, 8, STEP_NW>(a.tcfg, 1, a.dcap, d_n, d_box, d_cls, d_conf, a.tcap, a.trk_state, a.snap, a.snap_n,
                                        a.det2trk, 1, s, smem, 0xFEDCBA9876543210ull, stage, seq, 0,
                                        seq ? a.flags + 32 * (2 * s) : nullptr, (int)((unsigned)a.seq + 1u), (a.fence & 8) != 0, d_area);
        // (overlapped: the wave that keeps the complete rows has published the step counter itself, behind an s_waitcnt vmcnt(0) on its
        // record stores -- __syncthreads() compiles to s_waitcnt lgkmcnt(0) + s_barrier on this target and waits for no global store --
        // and written the snapshot rows after that; the successor may start before the wire table is written: the steps in flight have
        // wire buffers of their own)
        ck.mark(3);                   // tracker frame (thread 0's wave)
        if (a.wire) {                 // this stream's table in wire format (pack_tracks_kernel's row conversion)
            __syncthreads();          // the snapshot rows the bookkeeper wave wrote
            // frame = frame0 + the stream's detector frame count after this step (a captured graph -- fixed kernel arguments -- stamps
            // every replay with its own index)
            if (tid < a.tcap)
                wire_put(a.wire + (size_t)s * (AV_WIRE_HDR_BYTES + (size_t)a.tcap * AV_WIRE_ROW_BYTES), tid, a.snap_n[s], a.tcap,
                         a.snap + (size_t)s * a.tcap, a.stream0 + s, a.frame0 + fc_stage);
        }
    } else {
        const int s = blockIdx.x - a.S;
        StepClock ck;
        ck.start(a, 1);
        if (tid < 4) z_stage[tid] = a.z[(size_t)s * 4 + tid];
        if (seq && !seq_enter(a, 2 * s + 1, &go)) return;
        ck.mark(1);
        if (tid < 64) {
            double* rec = a.kf_state + (size_t)s * AV_KF_STATE_DOUBLES;
            fetch_record(rec, kf_stage, AV_KF_STATE_DOUBLES, 0, 64, seq);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            ck.mark(2);
            const double* z = z_stage;
            double *vs = a.vstate + (size_t)s * AV_VSTATE_DOUBLES, *ps = a.plan_state + (size_t)s * 4;
            const bool separable = kf_axis_body(a.kcfg, 1, z, nullptr, kf_stage, vs, ps, 0, tid);
            if (!separable && tid == 0) kf_dense_stream_lds(a.kcfg, 0, 1, z, nullptr, kf_stage, vs, ps);   // (LDS form: kf_dense.inc)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if (tid < AV_KF_STATE_DOUBLES) {
                if (seq)
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(rec) + tid, (unsigned long long)__double_as_longlong(kf_stage[tid]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    rec[tid] = kf_stage[tid];
            }
        }
        ck.mark(3);                   // Kalman step + record written
        if (seq && tid < 64) {        // the Kalman wave publishes its counter itself, behind the acknowledgement of its record stores
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            if (tid == 0) seq_leave(a, 2 * s + 1);
        }
        ck.mark(4);
        __syncthreads();              // the planner's start state (plan_state[s]) is in memory and visible to this workgroup
        plan_block<1, PW>(a.pp, s, a.S, a.plan_state, nullptr, 0, nullptr, 0, a.wp, a.cost, a.order, reinterpret_cast<double*>(smem));
        ck.mark(6);                   // planner (thread 0's wave)
    }
}

}  // namespace

// validates one step's arguments and fills its kernel arguments (lds: dynamic LDS bytes, pw: waves per workgroup)
static int hot_step_args(av_ctx* ctx, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg, int n_streams, int h,
                           int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                           double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n,
                           int32_t* det2trk, const double* z, double* kf_state, double* vstate, double* plan_state,
                           double* waypoints, double* cost, int32_t* order, void* wire, int stream0, int frame0, int32_t* seq_flags,
                           int seq, StepArgs& a, size_t& lds_out, int& pw_out, int depth = 1) {
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
    a = StepArgs{};
    a.S = n_streams, a.h = h, a.w = w, a.dcap = dcap, a.tcap = tcap;
    a.tcfg = *tcfg, a.kcfg = *kcfg;
    fill_params(ctx, a.pp);
    a.frame_count = frame_count, a.tab = (const SimRow*)ctx->d_simtab, a.cdf = ctx->d_cdf;
    a.det_n = det_n, a.det_box = det_box, a.det_cls = det_cls, a.det_conf = det_conf, a.det_status = det_status;
    a.trk_state = (unsigned char*)tracker_state, a.snap = snap, a.snap_n = snap_n, a.det2trk = det2trk;
    a.z = z, a.kf_state = kf_state, a.vstate = vstate, a.plan_state = plan_state;
    a.wp = waypoints, a.cost = cost, a.order = order;
    a.wire = (uint8_t*)wire, a.stream0 = stream0, a.frame0 = frame0;
    a.flags = seq_flags, a.seq = seq;
    const char* spe = seq_flags ? getenv("AVHOT_STEP_SPIN") : nullptr;
    a.spin = spe ? atoi(spe) : (1 << 22);
    const char* fe = seq_flags ? getenv("AVHOT_STEP_FENCE") : nullptr;
    a.fence = fe ? atoi(fe) : 0;     // (debug: 1 = full agent-scope acquire in every role, the form the comment above prices)
    // dynamic LDS: the larger of the tracker's (av_tracker_update's layout for one staged frame, eight replicas) and the planner's
    const int fc = 1;
    const size_t chunk_bytes = (size_t)((fc + 3) & ~3) * 4 + (size_t)fc * dcap * 16 + (((size_t)fc * dcap + 1) & ~size_t(1)) * 4 +
                               (size_t)fc * dcap * 16 + 16;
    const size_t rep_bytes = ((sizeof(Shared) + 63) & ~size_t(63)) + (size_t)tcap * sizeof(av_track_row);
    size_t lds_t = rep_bytes * STEP_NW + chunk_bytes + 2 * 576;
    lds_t = (lds_t + 15) & ~size_t(15);      // + the copy of the stream's header and rows the tracker role runs on
    a.stage_off = (int)lds_t;
    lds_t += HDR_INTS * 4 + (size_t)tcap * sizeof(av_track_row);
    // Waves per workgroup: sixteen (the planner's 21 trajectories in two rounds) unless that many launches in flight would not all be
    // resident -- every one of them may be waiting for the one before it, so `depth` launches of 2 S workgroups must fit on the
    // device together -- in which case eight (three rounds, twice the workgroups per CU).  AVHOT_STEP_PW=8|16 forces one.
    const char* pwe = getenv("AVHOT_STEP_PW");
    int pw = pwe && atoi(pwe) == 8 ? 8 : 16;
    size_t lds = 0;
    for (;;) {
        const size_t lds_p = plan_lds_doubles(1, ctx->n_points, ctx->n_cand, pw) * 8;
        lds = lds_t > lds_p ? lds_t : lds_p;
        if (depth <= 1) break;
        int per_cu = 0;
        if (pw == 16) AV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hot_step_kernel<16>, 16 * 64, lds));
        else AV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, hot_step_kernel<8>, 8 * 64, lds));
        if ((long long)depth * 2 * n_streams <= (long long)per_cu * ctx->n_cus) break;
        AV_REQUIRE(pw == 16 && !pwe, AV_EINVAL, "av_hot_step: %d launches of %d workgroups in flight do not fit the device (%d per CU x %d CUs)",
                   depth, 2 * n_streams, per_cu, ctx->n_cus);
        pw = 8;
    }
    // static __shared__ of the kernel (the Kalman bodies' arrays) counts against the same 64 KB
    static const size_t lds_static = [] {
        hipFuncAttributes fa{};
        return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(hot_step_kernel<16>)) == hipSuccess ? (size_t)fa.sharedSizeBytes : (size_t)16384;
    }();
    AV_REQUIRE(lds + lds_static <= 64 * 1024, AV_EINVAL, "av_hot_step: configuration needs %zu B of dynamic + %zu B of static LDS (limit 65536)",
               lds, lds_static);
    lds_out = lds, pw_out = pw;
    return AV_OK;
}

static int hot_step_go(const StepArgs& a, size_t lds, int pw, av_stream_t stream) {
    if (pw == 16) hipLaunchKernelGGL(hot_step_kernel<16>, dim3(2 * a.S), dim3(16 * 64), lds, as_stream(stream), a);
    else hipLaunchKernelGGL(hot_step_kernel<8>, dim3(2 * a.S), dim3(8 * 64), lds, as_stream(stream), a);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

static int hot_step_launch(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg, int n_streams, int h,
                           int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                           double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n,
                           int32_t* det2trk, const double* z, double* kf_state, double* vstate, double* plan_state,
                           double* waypoints, double* cost, int32_t* order, void* wire, int stream0, int frame0, int32_t* seq_flags,
                           int seq, int depth = 1) {
    StepArgs a;
    size_t lds;
    int pw;
    const int rc = hot_step_args(ctx, tcfg, kcfg, n_streams, h, w, dcap, tcap, frame_count, det_n, det_box, det_cls, det_conf, det_status,
                                 tracker_state, snap, snap_n, det2trk, z, kf_state, vstate, plan_state, waypoints, cost, order, wire, stream0,
                                 frame0, seq_flags, seq, a, lds, pw, depth);
    return rc != AV_OK ? rc : hot_step_go(a, lds, pw, stream);
}

extern "C" int av_hot_step(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg, int n_streams, int h,
                           int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                           double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n,
                           int32_t* det2trk, const double* z, double* kf_state, double* vstate, double* plan_state,
                           double* waypoints, double* cost, int32_t* order, void* wire, int stream0, int frame0) {
    return hot_step_launch(ctx, stream, tcfg, kcfg, n_streams, h, w, dcap, tcap, frame_count, det_n, det_box, det_cls, det_conf, det_status,
                           tracker_state, snap, snap_n, det2trk, z, kf_state, vstate, plan_state, waypoints, cost, order, wire, stream0,
                           frame0, nullptr, 0);
}

extern "C" int av_hot_step_seq(av_ctx* ctx, av_stream_t stream, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg, int n_streams, int h,
                               int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box, int32_t* det_cls,
                               double* det_conf, int32_t* det_status, void* tracker_state, av_track_row* snap, int32_t* snap_n,
                               int32_t* det2trk, const double* z, double* kf_state, double* vstate, double* plan_state,
                               double* waypoints, double* cost, int32_t* order, void* wire, int stream0, int frame0,
                               int32_t* seq_flags, int seq, int depth) {
    AV_REQUIRE(seq_flags, AV_EINVAL, "av_hot_step_seq: needs the sequence flags (AV_STEP_FLAG_INTS(n_streams) int32)");
    AV_REQUIRE(depth >= 2 && depth <= AV_STEP_MAX_DEPTH, AV_EINVAL, "av_hot_step_seq: depth %d not in [2, %d]", depth, AV_STEP_MAX_DEPTH);
    return hot_step_launch(ctx, stream, tcfg, kcfg, n_streams, h, w, dcap, tcap, frame_count, det_n, det_box, det_cls, det_conf, det_status,
                           tracker_state, snap, snap_n, det2trk, z, kf_state, vstate, plan_state, waypoints, cost, order, wire, stream0,
                           frame0, seq_flags, seq, depth);
}

extern "C" int av_hot_steps_seq(av_ctx* ctx, int depth, const av_stream_t* streams, const av_tracker_cfg* tcfg, const av_kf_cfg* kcfg,
                                int n_streams, int h, int w, int dcap, int tcap, int32_t* frame_count, int32_t* det_status,
                                void* tracker_state, double* kf_state, const av_step_set* sets, const double* z_steps, void* wire_steps,
                                int stream0, int frame0, int32_t* seq_flags, int seq0, int n_steps) {
    AV_REQUIRE(seq_flags && n_steps > 0 && sets && streams, AV_EINVAL, "av_hot_steps_seq: bad argument");
    AV_REQUIRE(depth >= 2 && depth <= AV_STEP_MAX_DEPTH, AV_EINVAL, "av_hot_steps_seq: depth %d not in [2, %d]", depth, AV_STEP_MAX_DEPTH);
    for (int k = 0; k < depth; ++k)
        for (int j = 0; j < k; ++j)
            AV_REQUIRE(streams[k] != streams[j], AV_EINVAL, "av_hot_steps_seq: the %d steps in flight need %d different HIP streams", depth, depth);
    StepArgs par[AV_STEP_MAX_DEPTH];
    size_t lds = 0;
    int pw = 16;
    for (int k = 0; k < depth; ++k) {
        const av_step_set& b = sets[k];
        const int rc = hot_step_args(ctx, tcfg, kcfg, n_streams, h, w, dcap, tcap, frame_count, b.det_n, b.det_box, b.det_cls, b.det_conf, det_status,
                                     tracker_state, b.snap, b.snap_n, b.det2trk, z_steps ? z_steps : b.z, kf_state, b.vstate, b.plan_state,
                                     b.waypoints, b.cost, b.order, wire_steps, stream0, frame0, seq_flags, seq0, par[k], lds, pw, depth);
        if (rc != AV_OK) return rc;
    }
    const size_t zb = (size_t)n_streams * 4, wb = (size_t)n_streams * (AV_WIRE_HDR_BYTES + (size_t)tcap * AV_WIRE_ROW_BYTES);
    for (int i = 0; i < n_steps; ++i) {
        const int q = (int)((unsigned)seq0 + (unsigned)i), k = (int)((unsigned)q % (unsigned)depth);      // (32-bit step numbers wrap)
        StepArgs& a = par[k];
        a.seq = q;
        if (z_steps) a.z = z_steps + (size_t)i * zb;
        if (wire_steps) a.wire = (uint8_t*)wire_steps + (size_t)i * wb;
        const int rc = hot_step_go(a, lds, pw, streams[k]);
        if (rc != AV_OK) return rc;
    }
    return AV_OK;
}
