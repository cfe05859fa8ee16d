// T1: rule-based maneuver tags from the Kalman output.
//
// Reference: ManeuverDetector.detect (src/tagging/maneuver_detector.py:105-262)
//   _detect_lateral_maneuver      :163-199  np.mean / np.std of the last 10 yaw rates; lane offset
//   _detect_longitudinal_maneuver :201-227  thresholds on speed and acceleration
//   _detect_turning_maneuver      :229-270  heading change over the last 15 states, +-360 normalisation
//
// The reference keeps deques of the last 30 states and looks at the newest 10 / 15 of them, so a frame's
// tags are a function of the current and the 14 previous vehicle states: thread = (stream, frame) over the
// av_kf_step output, with the 14 states before the window carried per stream.  np.mean/np.std of a 10-element
// list are reproduced in NumPy's pairwise order (8 partial sums combined as a tree, then the tail), so the
// confidences are the reference's bits.
#include "common.h"

namespace {

constexpr int MV_CARRY = 14;
constexpr double RAD2DEG = 57.29577951308232;      // 180.0 / pi as NumPy's npy_rad2deg forms it

// numpy add.reduce of 10 doubles (pairwise_sum, n < 128): r[0..7] as a tree, then + a[8] + a[9]
__device__ __forceinline__ double np_sum10(const double (&a)[10]) {
    double res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    res += a[8];
    res += a[9];
    return res;
}

__global__ void __launch_bounds__(256) maneuver_kernel(int n_streams, int n_frames, const double* __restrict__ vstate,
                                                       const double* __restrict__ lane_offset,
                                                       const double* __restrict__ state, av_maneuver_row* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n_streams * n_frames) return;
    const int s = (int)(i / n_frames), f = (int)(i - (long long)s * n_frames);
    const double* st = state + (size_t)s * AV_MANEUVER_STATE_DOUBLES;
    const double* vs = vstate + (size_t)s * n_frames * AV_VSTATE_DOUBLES;
    const long long seen = (long long)st[0] + f + 1;          // states in the reference's history incl. this one
    // state of relative frame j (j <= f; negative: from the carry, which holds frames -14..-1)
    auto yaw_at = [&](int j) { return j >= 0 ? vs[(size_t)j * AV_VSTATE_DOUBLES + 7] : st[1 + MV_CARRY + j]; };
    auto head_at = [&](int j) { return j >= 0 ? vs[(size_t)j * AV_VSTATE_DOUBLES + 4] : st[1 + 2 * MV_CARRY + j]; };
    const double* cur = vs + (size_t)f * AV_VSTATE_DOUBLES;
    const double heading = cur[4], speed = cur[5], acc = cur[6], yaw = cur[7];
    av_maneuver_row o;
    o.reserved = 0;
    o.timestamp = (double)(seen - 1) / 30.0;                    // :121  frame_count / 30.0
    o.speed_kmh = speed * 3.6, o.acceleration = acc, o.yaw_rate_deg = yaw * RAD2DEG;
    // ---- lateral (:163-199) ----------------------------------------------------------------------
    int lat = 0;
    double lat_c = 0.8;
    bool decided = false;
    if (seen >= 10) {
        double a[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) a[k] = yaw_at(f - 9 + k);
        const double mean = np_sum10(a) / 10.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const double d = a[k] - mean;
            a[k] = d * d;
        }
        const double sd = sqrt(np_sum10(a) / 10.0);
        const double avg_deg = mean * RAD2DEG;
        if (sd > 0.1) lat = 3, lat_c = fmin(0.9, sd * 5), decided = true;
        else if (avg_deg > 5.0) lat = 1, lat_c = fmin(0.9, fabs(avg_deg) / 20.0), decided = true;
        else if (avg_deg < -5.0) lat = 2, lat_c = fmin(0.9, fabs(avg_deg) / 20.0), decided = true;
    }
    if (!decided && lane_offset) {
        const double off = lane_offset[(size_t)s * n_frames + f];
        if (off == off && fabs(off) > 0.5) lat = off > 0 ? 1 : 2, lat_c = 0.6;            // NaN = no offset given
    }
    o.lateral = lat, o.lateral_confidence = lat_c;
    // ---- longitudinal (:201-227) -------------------------------------------------------------------
    int lon = 0;
    double lon_c = 0.8;
    if (speed < 0.5) lon = 4, lon_c = 0.95;
    else if (acc < -3.0) lon = 3, lon_c = fmin(0.95, fabs(acc) / 5.0);
    else if (acc < -1.0) lon = 2, lon_c = fmin(0.9, fabs(acc) / 3.0);
    else if (acc > 1.0) lon = 1, lon_c = fmin(0.9, acc / 3.0);
    o.longitudinal = lon, o.longitudinal_confidence = lon_c;
    // ---- turning (:229-270) ------------------------------------------------------------------------
    int trn = 0;
    double trn_c = 0.5;
    if (seen >= 15) {
        double hc = (heading - head_at(f - 14)) * RAD2DEG;
        while (hc > 180) hc -= 360;
        while (hc < -180) hc += 360;
        const double yd = yaw * RAD2DEG;
        if (fabs(hc) > 120) trn = 3, trn_c = 0.8;
        else if (hc > 60) trn = 1, trn_c = fmin(0.9, hc / 90);
        else if (hc < -60) trn = 2, trn_c = fmin(0.9, fabs(hc) / 90);
        else if (hc > 15) trn = 4, trn_c = fmin(0.8, hc / 45);
        else if (hc < -15) trn = 5, trn_c = fmin(0.8, fabs(hc) / 45);
        else if (fabs(yd) > 15.0) trn = yd > 0 ? 4 : 5, trn_c = 0.6;
        else trn = 0, trn_c = 0.8;
    }
    o.turning = trn, o.turning_confidence = trn_c;
    out[i] = o;
}

// advance the carry: the last 14 (yaw, heading) pairs and the number of states seen; one thread per stream
__global__ void maneuver_carry_kernel(int n_streams, int n_frames, const double* __restrict__ vstate, double* __restrict__ state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    double* st = state + (size_t)s * AV_MANEUVER_STATE_DOUBLES;
    const double* vs = vstate + (size_t)s * n_frames * AV_VSTATE_DOUBLES;
    double ny[MV_CARRY], nh[MV_CARRY];
    for (int k = 0; k < MV_CARRY; ++k) {
        const int j = n_frames - MV_CARRY + k;                   // relative frame that ends up in slot k
        ny[k] = j >= 0 ? vs[(size_t)j * AV_VSTATE_DOUBLES + 7] : st[1 + MV_CARRY + j];
        nh[k] = j >= 0 ? vs[(size_t)j * AV_VSTATE_DOUBLES + 4] : st[1 + 2 * MV_CARRY + j];
    }
    for (int k = 0; k < MV_CARRY; ++k) st[1 + k] = ny[k], st[1 + MV_CARRY + k] = nh[k];
    st[0] += (double)n_frames;
}

__global__ void maneuver_reset_kernel(int n, double* state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) state[i] = 0.0;
}

}  // namespace

extern "C" {

int av_maneuver_reset(av_ctx* ctx, av_stream_t stream, int n_streams, double* state) {
    AV_REQUIRE(ctx && state && n_streams > 0, AV_EINVAL, "av_maneuver_reset: bad argument");
    const int n = n_streams * AV_MANEUVER_STATE_DOUBLES;
    hipLaunchKernelGGL(maneuver_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), n, state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_maneuver_detect(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, const double* vstate,
                       const double* lane_offset, double* state, av_maneuver_row* out) {
    AV_REQUIRE(ctx && vstate && state && out, AV_EINVAL, "av_maneuver_detect: null argument");
    AV_REQUIRE(n_streams > 0 && n_frames > 0, AV_EINVAL, "av_maneuver_detect: n_streams/n_frames must be > 0");
    const long long n = (long long)n_streams * n_frames;
    hipLaunchKernelGGL(maneuver_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n_streams,
                       n_frames, vstate, lane_offset, state, out);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(maneuver_carry_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), n_streams,
                       n_frames, vstate, state);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"
