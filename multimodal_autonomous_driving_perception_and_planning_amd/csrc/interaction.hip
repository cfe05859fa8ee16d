// T2: rule-based interaction tags from the per-frame track tables.
//
// Reference: InteractionDetector.detect (src/tagging/interaction_detector.py:132-222)
//   _estimate_distance :224-248, _estimate_relative_speed :250-260, _calculate_ttc :262-268,
//   _analyze_interaction :270-372, _calculate_overall_risk :374-395, sort at :214.
//
// Mapping: lane = row of the tracker's snapshot table (tcap 64).  The only state that crosses frames is, per
// track, the ring of its last 30 centre abscissae (the cut-in rule looks at the oldest and the newest); it lives
// in LDS indexed by the row's history slot (stable while the track lives; a changed id in a slot means the slot
// was re-used and the ring restarts).  A confirmed track is returned in EVERY frame until it dies, so the ring at
// frame f is a function of the snapshot tables of frames f-29..f alone: a wave owns a chunk of IC frames of one
// stream and first replays the 29 frames before it (ring updates only), which makes the chunks independent --
// the window is spread over W/IC waves instead of one 1.8-us-per-frame chain.  Chunk 0 starts from the carried
// state, the last chunk leaves the next state in a scratch copy that a second launch moves into place (another
// chunk may still be reading the old one).  The next frame's rows are in flight while a frame is evaluated.  Per-track rules are element-wise; the frame summary is a handful of ballots and
// wave minima.  All arithmetic in the reference's operation order (Python float == IEEE double).
#include "common.h"

namespace {

constexpr int IH = 30;                       // history_length (:126)
constexpr int IC = 64;                       // frames per chunk

__host__ __device__ inline size_t ihalf_bytes(int tcap) { return 16 + (size_t)tcap * (8 + IH * 8); }
__host__ __device__ inline size_t istate_bytes(int tcap) { return 2 * ihalf_bytes(tcap); }      // state | scratch

__device__ __forceinline__ double wave_min_f64(double v) { return -wave_max(-v); }

__global__ void __launch_bounds__(64) interaction_kernel(av_interaction_cfg cfg, int n_frames, int tcap,
                                                         const av_track_row* __restrict__ snap, const int32_t* __restrict__ snap_n,
                                                         const double* __restrict__ vstate, const uint8_t* __restrict__ has_state,
                                                         const double* __restrict__ vy_in, unsigned char* __restrict__ state_all,
                                                         av_interaction_row* __restrict__ rows, av_interaction_summary* __restrict__ summ) {
    __shared__ double ring[64][IH];
    __shared__ int hid[64], hcnt[64];
    const int s = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x;
    const int c0 = chunk * IC, c1 = (c0 + IC) < n_frames ? (c0 + IC) : n_frames;
    unsigned char* st = state_all + (size_t)s * istate_bytes(tcap);
    const long long* hdr = reinterpret_cast<const long long*>(st);
    const int* g_id = reinterpret_cast<const int*>(st + 16);
    const int* g_cnt = g_id + tcap;
    const double* g_ring = reinterpret_cast<const double*>(st + 16 + (size_t)tcap * 8);
    const long long frame0 = hdr[0];
    const av_track_row* sp = snap + (size_t)s * n_frames * tcap;
    int r0 = c0 - (IH - 1);                       // first frame to replay
    if (r0 <= 0) {                                // the carried state covers everything before the window
        r0 = 0;
        hid[lane] = g_id[lane], hcnt[lane] = g_cnt[lane];
        for (int i = lane; i < 64 * IH; i += 64) ring[i / IH][i % IH] = g_ring[i];
    } else {
        hid[lane] = -1, hcnt[lane] = 0;
    }
    __syncthreads();
    // append this frame's centre to the track's ring; returns the length of its history
    auto append = [&](const av_track_row& g, bool on, double cx, double& start_x) -> int {
        if (!on) return 0;
        const int slot = g.slot & 63;
        int c = hid[slot] == g.id ? hcnt[slot] : 0;
        ring[slot][c % IH] = cx;
        c += 1;
        hid[slot] = g.id, hcnt[slot] = c;
        start_x = c <= IH ? ring[slot][0] : ring[slot][c % IH];
        return c < IH ? c : IH;
    };
    auto lds_order = [&]() {                       // one wave: order its LDS writes before the next frame's reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    av_track_row cur = sp[(size_t)r0 * tcap + lane];
    for (int f = r0; f < c0; ++f) {               // replay: history only
        const av_track_row g = cur;
        cur = sp[(size_t)(f + 1) * tcap + lane];   // f + 1 <= c0 < n_frames
        const bool on = lane < snap_n[(size_t)s * n_frames + f] && (g.flags & 1);
        double sx;
        append(g, on, (double)(g.x1 + g.x2) / 2, sx);
        lds_order();
    }
    const double w = (double)cfg.frame_w, h = (double)cfg.frame_h;
    const double w2 = w / 2, w4 = w / 4, w34 = 3 * w / 4;          // cfg.frame_w is an int: w/2 etc. as Python forms them
    for (int f = c0; f < c1; ++f) {
        const size_t sf = (size_t)s * n_frames + f;
        const av_track_row g = cur;
        if (f + 1 < c1) cur = sp[(size_t)(f + 1) * tcap + lane];              // prefetch
        const int n = snap_n[sf];
        const bool on = lane < n && (g.flags & 1);                             // a track the tracker returned
        const bool given = has_state ? has_state[sf] != 0 : true;
        const double ego = (vstate && given) ? vstate[sf * AV_VSTATE_DOUBLES + 5] : 10.0;   // :166
        const int kind = (g.cls >= 0 && g.cls < 16) ? cfg.class_kind[g.cls] : 0;
        // ---- per track ---------------------------------------------------------------------------------
        const int bh = g.y2 - g.y1;
        double dist = 50.0;                                                    // :233
        if (bh > 0) {
            const double yn = (double)g.y2 / h;
            const double base = 50.0 * (1 - yn) + 5.0;
            const double size = 100.0 / (double)(bh + 10);
            dist = fmax(2.0, fmin(100.0, (base + size) / 2));
        }
        const bool has_vel = g.hist_len >= 2;
        const double vy = vy_in ? vy_in[sf * tcap + lane] : (double)g.vy;
        const double rel = has_vel ? ego - vy : 0.0;                           // :252-260
        const bool has_ttc = rel > 0.1;
        const double ttc = has_ttc ? dist / rel : 0.0;
        const double cx = (double)(g.x1 + g.x2) / 2;
        // history: append, then the oldest / newest of the last IH entries
        double start_x = cx;
        const int hl = append(g, on, cx, start_x);
        int type = -1, risk = 0;
        double conf = 0.0, o_rel = rel, o_ttc = has_ttc ? ttc : __builtin_nan("");
        if (on) {
            if (dist < 3.0) {
                type = 9, conf = 0.9, risk = 3;
            } else if (kind == 1 && dist < 10.0) {
                if (fabs(cx - w2) < w4) type = 6, conf = 0.8, risk = dist < 8 ? 2 : 1;
                else type = 7, conf = 0.6, risk = 0, o_rel = 0.0, o_ttc = __builtin_nan("");
            } else if (kind == 2 && dist < 15) {
                type = 8, conf = 0.7, risk = dist < 8 ? 1 : 0, o_ttc = __builtin_nan("");
            } else if (kind == 3) {
                if (cx > w4 && cx < w34 && dist > 5.0 && dist < 30.0) {
                    risk = dist < 10 ? 1 : 0;
                    if (has_ttc && ttc != 0.0 && ttc < 3.0) risk = 2;            // `if ttc and ttc < TTC_WARNING`
                    type = 1, conf = 0.75;
                } else if (hl >= 10 && fabs(cx - w2) < fabs(start_x - w2) && dist < 15.0) {
                    type = 4, conf = 0.7, risk = 1, o_ttc = __builtin_nan("");
                }
            }
        }
        av_interaction_row o;
        o.type = type, o.risk = risk, o.agent_id = on ? g.id : -1, o.cls = g.cls;
        o.confidence = conf, o.distance = on ? dist : 0.0, o.relative_speed = type >= 0 ? o_rel : 0.0;
        o.ttc = type >= 0 ? o_ttc : __builtin_nan("");
        rows[sf * tcap + lane] = o;
        // ---- frame summary -----------------------------------------------------------------------------
        const unsigned long long m_on = __ballot(on);
        const int n_on = __popcll(m_on);
        av_interaction_summary q;
        q.agent_count = n_on;
        q.pedestrian_count = __popcll(__ballot(on && kind == 1));
        q.cyclist_count = __popcll(__ballot(on && kind == 2));
        q.vehicle_count = __popcll(__ballot(on && (kind == 3 || kind == 4)));
        const unsigned long long m_int = __ballot(type >= 0);
        q.n_interactions = __popcll(m_int);
        q.primary_type = -1, q.primary_row = -1, q.overall_risk = 0;
        q.timestamp = (double)(frame0 + f) / 30.0;
        if (n_on == 0) {
            q.closest_distance = __builtin_inf(), q.min_ttc = __builtin_nan("");      // untouched defaults (:143-151)
        } else {
            q.closest_distance = wave_min_f64(on ? dist : __builtin_inf());
            const double mt = wave_min_f64((on && has_ttc && ttc > 0) ? ttc : __builtin_inf());
            q.min_ttc = mt == __builtin_inf() ? __builtin_nan("") : mt;
            if (m_int) {
                // sort key (risk string, -confidence) reverse: 'medium' > 'low' > 'high' > 'critical', then the
                // SMALLEST confidence, then the first in track order
                const int rr = type >= 0 ? (risk == 1 ? 3 : risk == 0 ? 2 : risk == 2 ? 1 : 0) : -1;
                int best = rr;
                for (int off = 32; off > 0; off >>= 1) {
                    const int t = __shfl_xor(best, off, 64);
                    best = t > best ? t : best;
                }
                const bool c1 = type >= 0 && rr == best;
                const double mc = wave_min_f64(c1 ? conf : __builtin_inf());
                const unsigned long long m2 = __ballot(c1 && conf == mc);
                const int pl = __ffsll((long long)m2) - 1;
                q.primary_row = pl;
                q.primary_type = __shfl(type, pl, 64);
                int mr = type >= 0 ? risk : 0;
                for (int off = 32; off > 0; off >>= 1) {
                    const int t = __shfl_xor(mr, off, 64);
                    mr = t > mr ? t : mr;
                }
                q.overall_risk = (mt != 0.0 && mt < 1.5) ? 3 : mr;               // `if min_ttc and min_ttc < TTC_CRITICAL`
            }
        }
        if (lane == 0) summ[sf] = q;
        lds_order();
    }
    if (c1 != n_frames) return;
    // last chunk: the state the next window starts from, into the scratch half
    __syncthreads();
    unsigned char* sc = st + ihalf_bytes(tcap);
    int* o_id = reinterpret_cast<int*>(sc + 16);
    int* o_cnt = o_id + tcap;
    double* o_ring = reinterpret_cast<double*>(sc + 16 + (size_t)tcap * 8);
    o_id[lane] = hid[lane], o_cnt[lane] = hcnt[lane];
    for (int i = lane; i < 64 * IH; i += 64) o_ring[i] = ring[i / IH][i % IH];
    if (lane == 0) reinterpret_cast<long long*>(sc)[0] = frame0 + n_frames;
}

// scratch half -> state half (after every chunk of the window is done: stream order)
__global__ void interaction_commit_kernel(int n_streams, int tcap, unsigned char* state) {
    const size_t half8 = ihalf_bytes(tcap) / 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_streams * half8) return;
    const size_t s = i / half8, k = i - s * half8;
    long long* p = reinterpret_cast<long long*>(state + s * istate_bytes(tcap));
    p[k] = p[half8 + k];
}

__global__ void interaction_reset_kernel(size_t n8, long long* state) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n8) state[i] = 0;
}

__global__ void interaction_ids_kernel(int n_streams, int tcap, unsigned char* state) {     // ids start at -1 (no track)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams * tcap) return;
    const int s = i / tcap, k = i - s * tcap;
    reinterpret_cast<int*>(state + (size_t)s * istate_bytes(tcap) + 16)[k] = -1;
}

}  // namespace

extern "C" {

size_t av_interaction_state_bytes(int tcap) { return tcap > 0 ? istate_bytes(tcap) : 0; }

int av_interaction_reset(av_ctx* ctx, av_stream_t stream, int n_streams, int tcap, void* state) {
    AV_REQUIRE(ctx && state && n_streams > 0 && tcap == 64, AV_EINVAL, "av_interaction_reset: bad argument (tcap must be 64)");
    const size_t n8 = (size_t)n_streams * istate_bytes(tcap) / 8;
    hipLaunchKernelGGL(interaction_reset_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, as_stream(stream), n8,
                       (long long*)state);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(interaction_ids_kernel, dim3((n_streams * tcap + 255) / 256), dim3(256), 0, as_stream(stream), n_streams,
                       tcap, (unsigned char*)state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_interaction_detect(av_ctx* ctx, av_stream_t stream, const av_interaction_cfg* cfg, int n_streams, int n_frames,
                          int tcap, const av_track_row* snap, const int32_t* snap_n, const double* vstate,
                          const uint8_t* has_state, const double* vy, void* state, av_interaction_row* rows,
                          av_interaction_summary* summary) {
    AV_REQUIRE(ctx && cfg && snap && snap_n && state && rows && summary, AV_EINVAL, "av_interaction_detect: null argument");
    AV_REQUIRE(n_streams > 0 && n_frames > 0, AV_EINVAL, "av_interaction_detect: n_streams/n_frames must be > 0");
    AV_REQUIRE(tcap == 64, AV_EINVAL, "av_interaction_detect: tcap %d not supported (64 only)", tcap);
    AV_REQUIRE(cfg->frame_h > 0 && cfg->frame_w > 0, AV_EINVAL, "av_interaction_detect: bad frame shape");
    hipLaunchKernelGGL(interaction_kernel, dim3((n_frames + IC - 1) / IC, n_streams), dim3(64), 0, as_stream(stream), *cfg,
                       n_frames, tcap, snap, snap_n, vstate, has_state, vy, (unsigned char*)state, rows, summary);
    AV_LAUNCH_CHECK();
    const size_t n8 = (size_t)n_streams * ihalf_bytes(tcap) / 8;
    hipLaunchKernelGGL(interaction_commit_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, as_stream(stream), n_streams,
                       tcap, (unsigned char*)state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"
