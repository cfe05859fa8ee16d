// E1-E3: batched 6-state constant-acceleration Kalman estimator.
//
// Reference: VehicleStateEstimator (src/state_estimation/vehicle_state.py) over filterpy's
// KalmanFilter (third party, >=1.4.5, not vendored):
//   _init_kalman_filter :68-106  F = CA model(dt), H = [I4 0], R = r*I4, Q = diag(q,q,q,q,10q,10q)
//   predict   x = F x ; P = F P F^T + Q                       (filterpy, alpha = 1)
//   update    y = z - H x ; S = H P H^T + R ; K = P H^T S^-1 ; x += K y
//             P = (I-KH) P (I-KH)^T + K R K^T                 (filterpy, Joseph form)
//   _extract_state :158-198  speed/heading/acceleration/yaw-rate/uncertainties; it runs in
//             predict() AND update(), overwriting prev_heading/prev_speed both times
//
// Mapping: the filter of one stream is 42 doubles; one lane owns one stream and keeps x and P in
// registers for the whole window (state is read and written once per launch).  Products are
// written out with the known zeros/ones of F and H removed; a removed term is an exact 0, so the
// remaining sum, evaluated in ascending-k order without FMA contraction, equals the dense product.
#include "common.h"

namespace {

struct KfOut {
    double v[AV_VSTATE_DOUBLES];
};

struct Filter {
    double x[6];
    double P[6][6];
    double prev_heading, prev_speed, time;
};

__device__ __forceinline__ void extract(Filter& k, double dt, KfOut& o) {
    const double vx = k.x[2], vy = k.x[3];
    const double speed = sqrt(vx * vx + vy * vy);
    const double heading = speed > 0.1 ? atan2(vy, vx) : k.prev_heading;
    const double acc = dt > 0.0 ? (speed - k.prev_speed) / dt : 0.0;
    double dh = heading - k.prev_heading;
    const double pi = 3.141592653589793;
    if (dh > pi) dh -= 2.0 * pi;
    else if (dh < -pi) dh += 2.0 * pi;
    const double yaw = dt > 0.0 ? dh / dt : 0.0;
    o.v[0] = k.x[0], o.v[1] = k.x[1], o.v[2] = vx, o.v[3] = vy;
    o.v[4] = heading, o.v[5] = speed, o.v[6] = acc, o.v[7] = yaw, o.v[8] = k.time;
    o.v[9] = sqrt(k.P[0][0] + k.P[1][1]);
    o.v[10] = sqrt(k.P[2][2] + k.P[3][3]);
    o.v[11] = 0.0;
    k.prev_heading = heading;
    k.prev_speed = speed;
}

#define KF_FN(name) name
#define KF_UNROLL _Pragma("unroll")
#define KF_LOCAL
#define KF_INLINE __forceinline__
#include "kf_dense.inc"
#undef KF_FN
#undef KF_UNROLL
#undef KF_LOCAL
#undef KF_INLINE
#define KF_FN(name) name##_lds
#define KF_UNROLL _Pragma("unroll 1")
#define KF_LOCAL __shared__
#define KF_INLINE __forceinline__
#include "kf_dense.inc"
#undef KF_FN
#undef KF_UNROLL
#undef KF_LOCAL
#undef KF_INLINE

__global__ void __launch_bounds__(64) kf_kernel(av_kf_cfg cfg, int n_streams, int n_frames,
                                                const double* __restrict__ z, const uint8_t* __restrict__ mode,
                                                double* __restrict__ kf_state, double* __restrict__ out_state,
                                                double* __restrict__ plan_state, int only_flagged) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    if (only_flagged && kf_state[(size_t)s * AV_KF_STATE_DOUBLES + 45] == 0.0) return;       // handled by kf_axis_kernel
    kf_dense_stream(cfg, s, n_frames, z, mode, kf_state, out_state, plan_state);
}

// ---------------------------------------------------------------------------------------------------
// Fast path: axis-separable filter, one wave per stream.
//
// F, H, Q, R and P0 are block-diagonal with respect to the x axis (states 0,2,4; measurements 0,2) and
// the y axis (states 1,3,5; measurements 1,3), so every cross-axis entry of P stays an exact 0 and the
// 6-state filter is two independent 3-state / 2-measurement filters.  Lane 0 runs the x axis, lane 1
// the y axis (same instruction stream).  The sequential recursion only produces positions, velocities
// and covariance diagonals; the expensive part of _extract_state (sqrt, atan2, divides; twice per
// step) is evaluated afterwards for 64 frames at once, lane = frame.  The one sequential
// thread through it -- "heading keeps its previous value while speed <= 0.1" and prev_speed -- is
// resolved with a ballot + one cross-lane read.  A stream whose P is not separable (user-assigned P)
// is flagged in kf_state[45] and handled by the dense kernel above.
struct Axis {
    double x0, x1, x2;             // position, velocity, acceleration
    double p[3][3];
};

__device__ __forceinline__ void axis_predict(Axis& a, double dt, double h, double q) {
    const double nx0 = a.x0 + dt * a.x1 + h * a.x2;
    const double nx1 = a.x1 + dt * a.x2;
    a.x0 = nx0, a.x1 = nx1;
    double A[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A[0][c] = a.p[0][c] + dt * a.p[1][c] + h * a.p[2][c];
        A[1][c] = a.p[1][c] + dt * a.p[2][c];
        A[2][c] = a.p[2][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a.p[r][0] = A[r][0] + A[r][1] * dt + A[r][2] * h;
        a.p[r][1] = A[r][1] + A[r][2] * dt;
        a.p[r][2] = A[r][2];
    }
    a.p[0][0] += q, a.p[1][1] += q, a.p[2][2] += q * 10.0;
}

__device__ __forceinline__ void axis_update(Axis& a, double zp, double zv, double rr, double (&K)[3][2]) {
    const double y0 = zp - a.x0, y1 = zv - a.x1;
    const double s00 = a.p[0][0] + rr, s01 = a.p[0][1], s10 = a.p[1][0], s11 = a.p[1][1] + rr;
    const double rdet = 1.0 / (s00 * s11 - s01 * s10);
    const double i00 = s11 * rdet, i01 = -s01 * rdet, i10 = -s10 * rdet, i11 = s00 * rdet;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        K[r][0] = a.p[r][0] * i00 + a.p[r][1] * i10;
        K[r][1] = a.p[r][0] * i01 + a.p[r][1] * i11;
    }
    a.x0 = a.x0 + (K[0][0] * y0 + K[0][1] * y1);
    a.x1 = a.x1 + (K[1][0] * y0 + K[1][1] * y1);
    a.x2 = a.x2 + (K[2][0] * y0 + K[2][1] * y1);
    double A[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A[r][0] = ((r == 0) ? 1.0 : 0.0) - K[r][0];
        A[r][1] = ((r == 1) ? 1.0 : 0.0) - K[r][1];
    }
    double B[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = A[r][0] * a.p[0][c] + A[r][1] * a.p[1][c];
            if (r == 2) acc = acc + a.p[2][c];
            B[r][c] = acc;
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double acc = B[r][0] * A[c][0] + B[r][1] * A[c][1];
            if (c == 2) acc = acc + B[r][2];
            const double jo = (K[r][0] * rr) * K[c][0] + (K[r][1] * rr) * K[c][1];
            a.p[r][c] = acc + jo;
        }
}

constexpr int KF_BATCH = 64;
constexpr int RAW = 12;   // per frame: px py vx vy vxp vyp Ppos_x Ppos_y Pvel_x Pvel_y time mode

// one stream by one wave (a device function of the stream index: kf_axis_kernel runs it with s = blockIdx.x, the fused time-step
// kernel of step.hip in the first wave of a planner workgroup).  Returns false when the stream is flagged for the dense filter.
__device__ __forceinline__ bool kf_axis_body(const av_kf_cfg& cfg, int n_frames, const double* __restrict__ z,
                                             const uint8_t* __restrict__ mode, double* __restrict__ kf_state,
                                             double* __restrict__ out_state, double* __restrict__ plan_state, const int s, const int lane) {
    __shared__ __attribute__((aligned(16))) double zl[KF_BATCH][4];
    __shared__ __attribute__((aligned(16))) double raw[KF_BATCH][RAW];
    __shared__ int ml[KF_BATCH];
    double* st = kf_state + (size_t)s * AV_KF_STATE_DOUBLES;

    // separability check (uniform result): 18 cross-axis entries must be exact zeros
    bool bad = st[45] != 0.0;
    if (lane < 36) {
        const int r = lane / 6, c = lane - r * 6;
        if (((r ^ c) & 1) && st[6 + lane] != 0.0) bad = true;
    }
    if (__ballot(bad) != 0ull) {
        if (lane == 0) st[45] = 1.0;
        return false;
    }
    const int ax = lane & 1;                       // lanes >= 2 mirror lanes 0/1 (results unused)
    Axis a;
    a.x0 = st[ax], a.x1 = st[2 + ax], a.x2 = st[4 + ax];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a.p[r][c] = st[6 + (2 * r + ax) * 6 + (2 * c + ax)];
    double carry_h = st[42], carry_sp = st[43], time = st[44];
    double pp[3][3], Kc[3][2] = {{0, 0}, {0, 0}, {0, 0}};          // previous posterior P; gain of the last full update
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) pp[r][c] = a.p[r][c];
    bool steady = false;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const double dt = cfg.dt, h = 0.5 * (dt * dt), q = cfg.process_noise, rr = cfg.measurement_noise;

    for (int f0 = 0; f0 < n_frames; f0 += KF_BATCH) {
        const int nb = (n_frames - f0) < KF_BATCH ? (n_frames - f0) : KF_BATCH;
        const size_t sf0 = (size_t)s * n_frames + f0;
        int m_mine = 0;
        if (lane < nb) {
            const int m = mode ? (int)mode[sf0 + lane] : 1;
            m_mine = m;
            ml[lane] = m;
            double4 zz = make_double4(0.0, 0.0, 0.0, 0.0);
            if (z && (m == 1 || m == 3)) zz = reinterpret_cast<const double4*>(z)[sf0 + lane];
            *reinterpret_cast<double4*>(zl[lane]) = zz;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

        const unsigned long long ones = __ballot(m_mine == 1);   // frames of the batch that predict + update (scalar test in the loop)
        // ---- sequential recursion (lanes 0 and 1 carry the two axes) --------------------------------
        for (int fb = 0; fb < nb;) {
            if (steady) {
                // The covariance recursion does not see the measurements.  Once a predict+update frame has
                // reproduced the previous posterior P bit for bit, every further predict+update frame computes
                // the same prior, gain and posterior again: only the state moves (same expressions as below).
                // A loop of its own: the only values it carries from frame to frame are the state and the clock (inside
                // the general loop the compiler moved the whole filter, ~40 register pairs, around the back edge on every frame:
                // 935 cycles per frame, of which the arithmetic was a quarter).
                const double p00 = a.p[0][0], p11 = a.p[1][1];
                while (fb < nb && ((ones >> fb) & 1ull)) {
                    const double zp = zl[fb][ax], zv = zl[fb][2 + ax];
                    const double nx0 = a.x0 + dt * a.x1 + h * a.x2;
                    const double nx1 = a.x1 + dt * a.x2;
                    a.x0 = nx0, a.x1 = nx1;
                    time += dt;
                    const double vpred = a.x1;
                    const double y0 = zp - a.x0, y1 = zv - a.x1;
                    a.x0 = a.x0 + (Kc[0][0] * y0 + Kc[0][1] * y1);
                    a.x1 = a.x1 + (Kc[1][0] * y0 + Kc[1][1] * y1);
                    a.x2 = a.x2 + (Kc[2][0] * y0 + Kc[2][1] * y1);
                    if (lane < 2) {
                        double* w = raw[fb];
                        w[ax] = a.x0, w[2 + ax] = a.x1, w[4 + ax] = vpred;
                        w[6 + ax] = p00, w[8 + ax] = p11;
                        if (lane == 0) w[10] = time, w[11] = 1.0;
                    }
                    ++fb;
                }
                if (fb >= nb) break;
            }
            const int m = ml[fb];
            const double zp = zl[fb][ax], zv = zl[fb][2 + ax];
            double vpred = a.x1;
            {
                if (m != 3) {
                    axis_predict(a, dt, h, q);
                    time += dt;
                    vpred = a.x1;
                }
                if (m == 1 || m == 3) axis_update(a, zp, zv, rr, Kc);
                bool same = m == 1;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) same = same && a.p[r][c] == pp[r][c], pp[r][c] = a.p[r][c];
                steady = __ballot(same) == ~0ull;            // every lane mirrors one of the two axes
            }
            if (lane < 2) {
                double* w = raw[fb];
                w[ax] = a.x0, w[2 + ax] = a.x1, w[4 + ax] = vpred;
                w[6 + ax] = a.p[0][0], w[8 + ax] = a.p[1][1];
                if (lane == 0) w[10] = time, w[11] = (double)m;
            }
            ++fb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

        // ---- _extract_state for the whole batch, lane = frame ------------------------------------------
        const bool on = lane < nb;
        const double* w = raw[on ? lane : 0];
        const int m = (int)w[11];
        const double px = w[0], py = w[1], vx = w[2], vy = w[3], vxp = w[4], vyp = w[5];
        const bool has_pred = (m != 3), has_post = (m == 1 || m == 3);
        const double sp_p = sqrt(vxp * vxp + vyp * vyp);       // speed at the predict-time extract
        const double sp_q = sqrt(vx * vx + vy * vy);           // speed at the final extract (== sp_p for m 0/2)
        const bool val_p = has_pred && sp_p > 0.1, val_q = has_post && sp_q > 0.1;
        const double at_p = atan2(vyp, vxp), at_q = atan2(vy, vx);
        // heading this lane hands on if any of its extracts had speed > 0.1 (the last such one)
        const bool gives = on && (val_p || val_q);
        const double give_h = val_q ? at_q : at_p;
        const unsigned long long gmask = __ballot(gives);
        const unsigned long long below = gmask & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __clzll((long long)below) : 0;
        const double from_h = __shfl(give_h, src, 64);
        const double h_in = below ? from_h : carry_h;         // prev_heading entering this frame
        const double last_sp = has_post ? sp_q : sp_p;          // prev_speed this frame hands on
        double sp_in = __shfl_up(last_sp, 1, 64);
        if (lane == 0) sp_in = carry_sp;
        // first extract (after predict), then the final one
        const double h_p = has_pred ? (val_p ? at_p : h_in) : h_in;
        const double prev_h = has_pred ? h_p : h_in;
        const double prev_s = has_pred ? sp_p : sp_in;
        double heading, speed, ph, ps;
        if (m == 0) heading = h_p, speed = sp_p, ph = h_in, ps = sp_in;
        else if (m == 2) heading = (sp_p > 0.1 ? at_p : prev_h), speed = sp_p, ph = prev_h, ps = prev_s;
        else heading = (val_q ? at_q : prev_h), speed = sp_q, ph = prev_h, ps = prev_s;
        const double acc = dt > 0.0 ? (speed - ps) / dt : 0.0;
        double dh = heading - ph;
        const double pi = 3.141592653589793;
        if (dh > pi) dh -= 2.0 * pi;
        else if (dh < -pi) dh += 2.0 * pi;
        const double yaw = dt > 0.0 ? dh / dt : 0.0;
        if (on) {
            double2* dst = reinterpret_cast<double2*>(out_state + (sf0 + lane) * AV_VSTATE_DOUBLES);
            dst[0] = make_double2(px, py), dst[1] = make_double2(vx, vy), dst[2] = make_double2(heading, speed);
            dst[3] = make_double2(acc, yaw), dst[4] = make_double2(w[10], sqrt(w[6] + w[7]));
            dst[5] = make_double2(sqrt(w[8] + w[9]), 0.0);
            if (plan_state) reinterpret_cast<double4*>(plan_state)[sf0 + lane] = make_double4(px, py, heading, speed);
        }
        carry_h = __shfl(heading, nb - 1, 64);
        carry_sp = __shfl(speed, nb - 1, 64);
    }

    // ---- persist -----------------------------------------------------------------------------------------
    if (lane < 2) {
        st[ax] = a.x0, st[2 + ax] = a.x1, st[4 + ax] = a.x2;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) st[6 + (2 * r + ax) * 6 + (2 * c + ax)] = a.p[r][c];
        if (lane == 0) st[42] = carry_h, st[43] = carry_sp, st[44] = time;
    }
    return true;
}

// DENSE: a stream that turns out not to be separable is finished here, by lane 0 with the dense filter's LDS form (kf_dense.inc),
// instead of by a second launch of kf_kernel -- the frame-by-frame calls (one or two frames per launch), where that second launch
// was 5 us of every call.  Windows keep the register form in its own launch.
template <bool DENSE>
__global__ void __launch_bounds__(64) kf_axis_kernel(av_kf_cfg cfg, int n_frames, const double* __restrict__ z,
                                                     const uint8_t* __restrict__ mode, double* __restrict__ kf_state,
                                                     double* __restrict__ out_state, double* __restrict__ plan_state) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of throughput kernels on the same SIMD
    const bool separable = kf_axis_body(cfg, n_frames, z, mode, kf_state, out_state, plan_state, blockIdx.x, threadIdx.x);
    if (DENSE && !separable && threadIdx.x == 0) kf_dense_stream_lds(cfg, blockIdx.x, n_frames, z, mode, kf_state, out_state, plan_state);
}

__global__ void kf_reset_kernel(int n_streams, double* kf_state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    double* st = kf_state + (size_t)s * AV_KF_STATE_DOUBLES;
    for (int i = 0; i < AV_KF_STATE_DOUBLES; ++i) st[i] = 0.0;
    for (int i = 0; i < 6; ++i) st[6 + i * 7] = 10.0;       // P0 = 10 I (vehicle_state.py:101)
}

}  // namespace

#ifndef AVHOT_DEVICE_ONLY      // (step.hip includes this file for its device code only)

extern "C" {

int av_kf_reset(av_ctx* ctx, av_stream_t stream, int n_streams, double* kf_state) {
    AV_REQUIRE(ctx && kf_state && n_streams > 0, AV_EINVAL, "av_kf_reset: bad argument");
    hipLaunchKernelGGL(kf_reset_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), n_streams, kf_state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_kf_step(av_ctx* ctx, av_stream_t stream, const av_kf_cfg* cfg, int n_streams, int n_frames, const double* z,
               const uint8_t* mode, double* kf_state, double* out_state, double* plan_state) {
    AV_REQUIRE(ctx && cfg && kf_state && out_state, AV_EINVAL, "av_kf_step: null argument");
    AV_REQUIRE(z || mode, AV_EINVAL, "av_kf_step: z may only be NULL when every mode is 0 or 2");
    AV_REQUIRE(n_streams > 0 && n_frames > 0, AV_EINVAL, "av_kf_step: n_streams/n_frames must be > 0");
    // separable streams: one wave each (kf_axis_kernel); the rest (flagged in kf_state[45]): dense kernel
    if (n_frames <= 2) {
        hipLaunchKernelGGL(kf_axis_kernel<true>, dim3(n_streams), dim3(64), 0, as_stream(stream), *cfg, n_frames, z, mode, kf_state,
                           out_state, plan_state);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    hipLaunchKernelGGL(kf_axis_kernel<false>, dim3(n_streams), dim3(64), 0, as_stream(stream), *cfg, n_frames, z, mode,
                       kf_state, out_state, plan_state);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(kf_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), *cfg, n_streams,
                       n_frames, z, mode, kf_state, out_state, plan_state, 1);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"

#endif  // AVHOT_DEVICE_ONLY
