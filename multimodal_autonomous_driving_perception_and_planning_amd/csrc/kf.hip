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

__device__ __forceinline__ void predict(Filter& k, double dt, double hdt2, double q) {
    // x = F x
    double x[6];
    x[0] = k.x[0] + dt * k.x[2] + hdt2 * k.x[4];
    x[1] = k.x[1] + dt * k.x[3] + hdt2 * k.x[5];
    x[2] = k.x[2] + dt * k.x[4];
    x[3] = k.x[3] + dt * k.x[5];
    x[4] = k.x[4];
    x[5] = k.x[5];
#pragma unroll
    for (int i = 0; i < 6; ++i) k.x[i] = x[i];
    // A = F P
    double A[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        A[0][c] = k.P[0][c] + dt * k.P[2][c] + hdt2 * k.P[4][c];
        A[1][c] = k.P[1][c] + dt * k.P[3][c] + hdt2 * k.P[5][c];
        A[2][c] = k.P[2][c] + dt * k.P[4][c];
        A[3][c] = k.P[3][c] + dt * k.P[5][c];
        A[4][c] = k.P[4][c];
        A[5][c] = k.P[5][c];
    }
    // P = A F^T + Q
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        k.P[r][0] = A[r][0] + A[r][2] * dt + A[r][4] * hdt2;
        k.P[r][1] = A[r][1] + A[r][3] * dt + A[r][5] * hdt2;
        k.P[r][2] = A[r][2] + A[r][4] * dt;
        k.P[r][3] = A[r][3] + A[r][5] * dt;
        k.P[r][4] = A[r][4];
        k.P[r][5] = A[r][5];
    }
    k.P[0][0] += q, k.P[1][1] += q, k.P[2][2] += q, k.P[3][3] += q;
    k.P[4][4] += q * 10.0, k.P[5][5] += q * 10.0;
    k.time += dt;
}

// inverse of a 4x4 by Gauss-Jordan elimination with partial (row) pivoting, fully unrolled
__device__ __forceinline__ void inv4(const double S[4][4], double X[4][4]) {
    double a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][c] = S[r][c], a[r][4 + c] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // pivot: first row i >= k with the largest |a[i][k]|
        int p = k;
        double best = fabs(a[k][k]);
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double v = fabs(a[i][k]);
            if (v > best) best = v, p = i;
        }
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const bool sw = (p == i);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double u = a[k][c], w = a[i][c];
                a[k][c] = sw ? w : u;
                a[i][c] = sw ? u : w;
            }
        }
        const double piv = a[k][k];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[k][c] = a[k][c] / piv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == k) continue;
            const double m = a[i][k];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[i][c] = a[i][c] - m * a[k][c];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) X[r][c] = a[r][4 + c];
}

__device__ __forceinline__ void update(Filter& k, const double z[4], double rr) {
    double y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = z[i] - k.x[i];
    double S[4][4], SI[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i][j] = k.P[i][j] + ((i == j) ? rr : 0.0);
    inv4(S, SI);
    double K[6][4];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = k.P[r][0] * SI[0][j];
#pragma unroll
            for (int m = 1; m < 4; ++m) acc = acc + k.P[r][m] * SI[m][j];
            K[r][j] = acc;
        }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double acc = K[r][0] * y[0];
#pragma unroll
        for (int m = 1; m < 4; ++m) acc = acc + K[r][m] * y[m];
        k.x[r] = k.x[r] + acc;
    }
    // A = I - K H  (columns 0..3 = delta - K, columns 4,5 = delta)
    double A[6][4];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) A[r][c] = ((r == c) ? 1.0 : 0.0) - K[r][c];
    // B = A P
    double B[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double acc = A[r][0] * k.P[0][c];
#pragma unroll
            for (int m = 1; m < 4; ++m) acc = acc + A[r][m] * k.P[m][c];
            if (r >= 4) acc = acc + k.P[r][c];
            B[r][c] = acc;
        }
    // P = B A^T + (K R) K^T
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double acc = B[r][0] * A[c][0];
#pragma unroll
            for (int m = 1; m < 4; ++m) acc = acc + B[r][m] * A[c][m];
            if (c >= 4) acc = acc + B[r][c];
            double jo = (K[r][0] * rr) * K[c][0];
#pragma unroll
            for (int m = 1; m < 4; ++m) jo = jo + (K[r][m] * rr) * K[c][m];
            k.P[r][c] = acc + jo;
        }
}

__global__ void __launch_bounds__(64) kf_kernel(av_kf_cfg cfg, int n_streams, int n_frames,
                                                const double* __restrict__ z, const uint8_t* __restrict__ mode,
                                                double* __restrict__ kf_state, double* __restrict__ out_state,
                                                double* __restrict__ plan_state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    Filter k;
    double* st = kf_state + (size_t)s * AV_KF_STATE_DOUBLES;
#pragma unroll
    for (int i = 0; i < 6; ++i) k.x[i] = st[i];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) k.P[r][c] = st[6 + r * 6 + c];
    k.prev_heading = st[42], k.prev_speed = st[43], k.time = st[44];
    const double dt = cfg.dt, hdt2 = 0.5 * (dt * dt), q = cfg.process_noise, rr = cfg.measurement_noise;

    for (int f = 0; f < n_frames; ++f) {
        const size_t sf = (size_t)s * n_frames + f;
        const int m = mode ? (int)mode[sf] : 1;
        KfOut o;
        if (m != 3) {
            predict(k, dt, hdt2, q);
            extract(k, dt, o);
        }
        if (m == 1 || m == 3) {
            double zz[4];
            const double4 zv = reinterpret_cast<const double4*>(z)[sf];
            zz[0] = zv.x, zz[1] = zv.y, zz[2] = zv.z, zz[3] = zv.w;
            update(k, zz, rr);
            extract(k, dt, o);
        } else if (m == 2) {
            extract(k, dt, o);
        }
        double* dst = out_state + sf * AV_VSTATE_DOUBLES;
#pragma unroll
        for (int i = 0; i < AV_VSTATE_DOUBLES; ++i) dst[i] = o.v[i];
        if (plan_state) reinterpret_cast<double4*>(plan_state)[sf] = make_double4(o.v[0], o.v[1], o.v[4], o.v[5]);
    }

#pragma unroll
    for (int i = 0; i < 6; ++i) st[i] = k.x[i];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) st[6 + r * 6 + c] = k.P[r][c];
    st[42] = k.prev_heading, st[43] = k.prev_speed, st[44] = k.time;
}

__global__ void kf_reset_kernel(int n_streams, double* kf_state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    double* st = kf_state + (size_t)s * AV_KF_STATE_DOUBLES;
    for (int i = 0; i < AV_KF_STATE_DOUBLES; ++i) st[i] = 0.0;
    for (int i = 0; i < 6; ++i) st[6 + i * 7] = 10.0;       // P0 = 10 I (vehicle_state.py:101)
}

}  // namespace

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
    hipLaunchKernelGGL(kf_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), *cfg, n_streams,
                       n_frames, z, mode, kf_state, out_state, plan_state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"
