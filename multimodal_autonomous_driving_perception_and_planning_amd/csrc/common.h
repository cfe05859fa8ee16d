// Internal helpers shared by the libavhot.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "avhot.h"

struct PlannerTables;

struct av_ctx {
    int device = -1;
    int n_cus = 256;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // planner
    bool planner_ready = false;
    av_planner_cfg pcfg{};
    int n_points = 0, n_lat = 0, n_cand = 0;
    double* d_ptab = nullptr;          // device: t[n] | alpha[n] | q[n] | dtdiff[n] | lat[n_lat]
    // simulated detector: class cdf (8 doubles); the draws of the 1000 possible per-frame seeds (simdet.hip)
    double* d_cdf = nullptr;
    void* d_simtab = nullptr;
    std::vector<hipGraphExec_t> graphs;
    // lane / yolo sub-contexts are attached by their own translation units
    void* lane = nullptr;
    void* yolo = nullptr;
};

void av_set_error(const char* fmt, ...);

#define AV_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            av_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return AV_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define AV_REQUIRE(cond, code, ...)   \
    do {                              \
        if (!(cond)) {                \
            av_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define AV_LAUNCH_CHECK()                                                                \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            av_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return AV_EHIP;                                                              \
        }                                                                                \
    } while (0)

static inline hipStream_t as_stream(av_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- wave64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// DPP moves (no LDS round trip).  ctrl: quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142, row_bcast:31 0x143, wave_shr:1 0x138 (lane i <- lane i-1), wave_shl:1 0x130.
template <int CTRL, int ROW_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, BOUND);
    const int nhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, BOUND);
    return __hiloint2double(nhi, nlo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Deterministic wave-wide sum (fixed tree), result uniform in every lane.
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov_f64<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);          // row_half_mirror
    v += dpp_mov_f64<0x140>(v);          // row_mirror
    v += dpp_mov_f64<0x142, 0xA>(v);     // row_bcast:15 -> rows 1,3
    v += dpp_mov_f64<0x143, 0xC>(v);     // row_bcast:31 -> rows 2,3
    return readlane_f64(v, 63);
}

// DPP move whose unwritten lanes (row_mask) keep their own value -- neutral for max/min.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_keep_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    const int nhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(nhi, nlo);
}
// v_max_f64; operands are never NaN here, so this is the plain maximum
__device__ __forceinline__ double fmax_nn(double a, double b) { return __builtin_fmax(a, b); }
// Wave-wide maximum (no NaNs expected), result uniform in every lane.
__device__ __forceinline__ double wave_max(double v) {
    v = fmax_nn(v, dpp_mov_keep_f64<0xB1>(v));
    v = fmax_nn(v, dpp_mov_keep_f64<0x4E>(v));
    v = fmax_nn(v, dpp_mov_keep_f64<0x141>(v));
    v = fmax_nn(v, dpp_mov_keep_f64<0x140>(v));
    v = fmax_nn(v, dpp_mov_keep_f64<0x142, 0xA>(v));
    v = fmax_nn(v, dpp_mov_keep_f64<0x143, 0xC>(v));
    return readlane_f64(v, 63);
}
// Unsigned 32-bit wave maximum: one VOP2 with a DPP operand per step.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    // old = 0 is the identity of an unsigned max (also for the lanes a partial row mask leaves unwritten),
    // which lets the DPP operand fold into v_max_u32_dpp
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_u32<0xB1>(v);
    v = dpp_max_u32<0x4E>(v);
    v = dpp_max_u32<0x141>(v);
    v = dpp_max_u32<0x140>(v);
    v = dpp_max_u32<0x142, 0xA>(v);
    v = dpp_max_u32<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Bitwise OR over the wave (0 is the identity, so the DPP operand folds like the max above).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_or_u32(unsigned v) {
    return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v = dpp_or_u32<0xB1>(v);
    v = dpp_or_u32<0x4E>(v);
    v = dpp_or_u32<0x141>(v);
    v = dpp_or_u32<0x140>(v);
    v = dpp_or_u32<0x142, 0xA>(v);
    v = dpp_or_u32<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Lanes holding the maximum of v over the `valid` lanes, v >= +0.0 there (never NaN): non-negative
// doubles order like their (hi, lo) words as unsigned integers, so two u32 reductions replace the
// f64 compare/select chain.  Returns 0 when no lane is valid.
__device__ __forceinline__ unsigned long long wave_argmax_nonneg(double v, bool valid) {
    const unsigned hi = valid ? (unsigned)__double2hiint(v) : 0u, lo = (unsigned)__double2loint(v);
    const unsigned mh = wave_max_u32(hi);
    const bool top = valid && hi == mh;
    const unsigned long long bt = __ballot(top);
    if (__popcll(bt) <= 1) return bt;              // the high words already single out the maximum
    const unsigned ml = wave_max_u32(top ? lo : 0u);
    return __ballot(top && lo == ml);
}
// ---- track-table wire format (include/avhot.h: av_wire_hdr / av_wire_row), the one device statement of it: used by
// pack_tracks_kernel (exchange.hip) and by the one-launch step (step.hip) -------------------------------------------
__device__ __forceinline__ av_wire_row wire_row_from(const av_track_row& in, bool live) {
    av_wire_row o;
    if (live) {
        o.id = in.id;
        o.x1 = (int16_t)in.x1, o.y1 = (int16_t)in.y1, o.x2 = (int16_t)in.x2, o.y2 = (int16_t)in.y2;
        o.age = in.age, o.hits = in.hits;
        o.misses = (uint16_t)(in.misses > 65535 ? 65535 : in.misses);
        o.cls = (uint8_t)in.cls, o.flags = (uint8_t)in.flags;
        o.conf = (float)in.conf;
        // centre velocities are differences of half-integers: 2*v is an exact integer
        o.vx2 = (int16_t)(in.vx * 2.0f), o.vy2 = (int16_t)(in.vy * 2.0f);
    } else {
        o.id = 0, o.x1 = o.y1 = o.x2 = o.y2 = 0, o.age = o.hits = 0, o.misses = 0, o.cls = 0, o.flags = 0, o.conf = 0.0f;
        o.vx2 = o.vy2 = 0;
    }
    return o;
}
// row r of one table: `rows` = the table's tcap tracker rows, n its live count, dst the table's first wire byte
__device__ __forceinline__ void wire_put(uint8_t* dst, int r, int n, int tcap, const av_track_row* rows, int stream, int frame) {
    if (r == 0) {
        av_wire_hdr h;
        h.n_rows = n < tcap ? n : tcap, h.stream = stream, h.frame = frame, h.reserved = 0;
        *reinterpret_cast<av_wire_hdr*>(dst) = h;
    }
    av_track_row in{};
    if (r < n) in = rows[r];
    reinterpret_cast<av_wire_row*>(dst + AV_WIRE_HDR_BYTES)[r] = wire_row_from(in, r < n);
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
