// D1: simulated detector on device.
//
// Reference: ObjectDetector._detect_simulated (src/perception/detector.py:125-169).  Every frame
// the reference reseeds NumPy's global *legacy* MT19937 stream with frame_count % 1000 (:134) and
// then draws: randint(3,8); per vehicle uniform(0.3,1.0), randint(-10,10), randint(-5,5),
// choice(8, p=class_weights), uniform(0.75,0.98).  The legacy stream is frozen by NumPy policy:
//   seed(int)      -> init_genrand: mt[0]=s, mt[i] = 1812433253*(mt[i-1]^(mt[i-1]>>30)) + i
//   randint(lo,hi) -> masked rejection on one 32-bit word: (next & mask) until <= hi-1-lo
//   random_sample  -> (a>>5, b>>6): (a*2^26 + b) / 2^53
//   uniform(lo,hi) -> lo + (hi-lo)*random_sample   (mul then add, no FMA)
//   choice(p)      -> cdf = cumsum(p)/cumsum(p)[-1]; count(cdf <= random_sample)
//
// A frame consumes ~70 words, far fewer than 227, so the first
// twist never reads a regenerated word: output k = temper(mt[k+397] ^ twist(mt[k], mt[k+1])) with
// all three taken from the seeding recurrence.  Two copies of that recurrence (at k and k+397)
// run in lock-step in registers; no 624-word state array exists.
#include "common.h"

namespace {

struct Mt {
    uint32_t a0, a1, b;   // mt[k], mt[k+1], mt[k+397]
    uint32_t k;
    uint32_t overflow;
};

__device__ __forceinline__ uint32_t seed_step(uint32_t prev, uint32_t i) {
    return 1812433253u * (prev ^ (prev >> 30)) + i;
}

__device__ __forceinline__ void mt_seed(Mt& m, uint32_t seed) {
    m.a0 = seed;
    m.a1 = seed_step(seed, 1);
    uint32_t b = m.a1;
    for (uint32_t i = 2; i <= 397; ++i) b = seed_step(b, i);
    m.b = b;
    m.k = 0;
    m.overflow = 0;
}

__device__ __forceinline__ uint32_t mt_next(Mt& m) {
    if (m.k >= 227u) m.overflow = 1;                      // would need regenerated words
    uint32_t y = (m.a0 & 0x80000000u) | (m.a1 & 0x7fffffffu);
    uint32_t v = m.b ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    // advance both recurrences
    m.a0 = m.a1;
    m.a1 = seed_step(m.a1, m.k + 2);
    m.b = seed_step(m.b, m.k + 398);
    m.k += 1;
    // tempering
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}

__device__ __forceinline__ double mt_double(Mt& m) {
    uint32_t a = mt_next(m) >> 5, b = mt_next(m) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

// RandomState.randint(lo, hi): values in [lo, hi-1]
__device__ __forceinline__ int mt_randint(Mt& m, int lo, int hi) {
    uint32_t rng = (uint32_t)(hi - 1 - lo);
    if (rng == 0) return lo;
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t val;
    int guard = 0;
    do {
        val = mt_next(m) & mask;
    } while (val > rng && ++guard < 256);
    return lo + (int)val;
}

__device__ __forceinline__ double mt_uniform(Mt& m, double lo, double range) {
    return lo + range * mt_double(m);     // library is built with -ffp-contract=off
}

// The draws of a frame depend on its seed (frame_count % 1000) only, the boxes also on the frame count itself and on
// the frame size.  So the MT19937 work -- seeding alone is a 397-step dependent chain, ~20 us of latency however few
// frames a launch holds -- is done ONCE per context for the 1000 possible seeds (simdet_table_kernel, at av_ctx_create),
// and a frame is a table row plus the box arithmetic.
struct SimVehicle {
    double depth, u, conf;      // uniform(0.3, 1.0), the class draw, uniform(0.75, 0.98)
    int32_t r1, r2;             // randint(-10, 10), randint(-5, 5)
};
struct SimRow {
    int32_t n, overflow;
    SimVehicle v[7];
};
static_assert(sizeof(SimRow) == 8 + 7 * 32, "SimRow layout");

__global__ void __launch_bounds__(64) simdet_table_kernel(SimRow* __restrict__ tab) {
    const int seed = blockIdx.x * blockDim.x + threadIdx.x;
    if (seed >= 1000) return;
    Mt m;
    mt_seed(m, (uint32_t)seed);
    SimRow r;
    r.n = mt_randint(m, 3, 8);
    for (int i = 0; i < 7; ++i) {
        SimVehicle v{};
        if (i < r.n) {                                  // draw order per vehicle: uniform, randint, randint, choice, uniform
            v.depth = mt_uniform(m, 0.3, 1.0 - 0.3);
            v.r1 = mt_randint(m, -10, 10);
            v.r2 = mt_randint(m, -5, 5);
            v.u = mt_double(m);
            v.conf = mt_uniform(m, 0.75, 0.98 - 0.75);
        }
        r.v[i] = v;
    }
    r.overflow = (int32_t)m.overflow;
    tab[seed] = r;
}

// thread = (stream, frame).  ADVANCE: the launch holds one frame per stream, so the thread that read the counter
// also writes it back (no second launch).
// one frame of one stream (a device function: the fused time-step kernel of step.hip calls it for its stream's single frame)
template <bool ADVANCE>
__device__ __forceinline__ void simdet_frame(long long gid, int s, int f, int h, int w, int dcap,
                                             int32_t* __restrict__ frame_count, const SimRow* __restrict__ tab,
                                             const double* __restrict__ cdf, int32_t* __restrict__ det_n,
                                             int32_t* __restrict__ det_box, int32_t* __restrict__ det_cls,
                                             double* __restrict__ det_conf, int32_t* __restrict__ status) {
    const int fc = frame_count[s] + f + 1;                 // detector.py:96 increments before use
    if (ADVANCE) frame_count[s] = fc;
    const SimRow& row = tab[((fc % 1000) + 1000) % 1000];
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = cdf[i];
    int n = row.n;
    if (n > dcap) n = dcap;
    const double t = (double)fc * 0.02;
    const size_t base = (size_t)gid * dcap;
    for (int i = 0; i < n; ++i) {
        const SimVehicle v = row.v[i];
        const double depth = v.depth;
        const int bw = (int)(80.0 * depth + 40.0);
        const int bh = (int)(60.0 * depth + 30.0);
        const int wob = (int)(50.0 * sin(t + (double)i));
        const int mod = w - bw;
        int xb = (i * 150 + wob) % mod;
        if (xb != 0 && ((xb < 0) != (mod < 0))) xb += mod;           // Python floor-mod
        const double h04 = (double)h * 0.4;
        const int yb = (int)(h04 + h04 * depth);
        int x1 = xb + v.r1;
        x1 = x1 > 0 ? x1 : 0;
        int y1 = yb + v.r2;
        y1 = y1 > 0 ? y1 : 0;
        const int x2 = (x1 + bw) < w ? (x1 + bw) : w;
        const int y2 = (y1 + bh) < h ? (y1 + bh) : h;
        int cls = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) cls += (c[q] <= v.u) ? 1 : 0;    // searchsorted(side='right')
        int32_t* b = det_box + (base + i) * 4;
        b[0] = x1, b[1] = y1, b[2] = x2, b[3] = y2;
        det_cls[base + i] = cls;
        det_conf[base + i] = v.conf;
    }
    for (int i = n; i < dcap; ++i) {
        int32_t* b = det_box + (base + i) * 4;
        b[0] = b[1] = b[2] = b[3] = 0;
        det_cls[base + i] = 0;
        det_conf[base + i] = 0.0;
    }
    det_n[gid] = n;
    if (row.overflow && status) atomicOr(&status[s], 1);
}

template <bool ADVANCE>
__global__ void __launch_bounds__(64) simdet_kernel(int n_streams, int n_frames, int h, int w, int dcap,
                                                    int32_t* __restrict__ frame_count, const SimRow* __restrict__ tab,
                                                    const double* __restrict__ cdf, int32_t* __restrict__ det_n,
                                                    int32_t* __restrict__ det_box, int32_t* __restrict__ det_cls,
                                                    double* __restrict__ det_conf, int32_t* __restrict__ status) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n_streams * n_frames) return;
    simdet_frame<ADVANCE>(gid, (int)(gid / n_frames), (int)(gid % n_frames), h, w, dcap, frame_count, tab, cdf, det_n, det_box, det_cls,
                          det_conf, status);
}

__global__ void advance_counter_kernel(int n_streams, int n_frames, int32_t* frame_count) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_streams) frame_count[s] += n_frames;
}

}  // namespace

#ifndef AVHOT_DEVICE_ONLY      // (step.hip includes this file for its device code only)

// called by av_ctx_create / av_ctx_destroy (ctx.hip)
int av_simdet_ctx_init(av_ctx* ctx) {
    AV_HIP(hipMalloc(&ctx->d_simtab, 1000 * sizeof(SimRow)));
    hipLaunchKernelGGL(simdet_table_kernel, dim3(16), dim3(64), 0, nullptr, (SimRow*)ctx->d_simtab);
    AV_LAUNCH_CHECK();
    AV_HIP(hipDeviceSynchronize());
    return AV_OK;
}

extern "C" int av_simdet_generate(av_ctx* ctx, av_stream_t stream, int n_streams, int n_frames, int h, int w,
                                  int dcap, int32_t* frame_count, int32_t* det_n, int32_t* det_box,
                                  int32_t* det_cls, double* det_conf, int32_t* status) {
    AV_REQUIRE(ctx && frame_count && det_n && det_box && det_cls && det_conf, AV_EINVAL,
               "av_simdet_generate: null argument");
    AV_REQUIRE(n_streams > 0 && n_frames > 0, AV_EINVAL, "av_simdet_generate: n_streams/n_frames must be > 0");
    AV_REQUIRE(dcap >= 7 && dcap <= 64, AV_EINVAL, "av_simdet_generate: dcap %d not in [7,64]", dcap);
    AV_REQUIRE(h > 0 && w > 121, AV_EINVAL, "av_simdet_generate: frame %dx%d too small (w - box_w must stay > 0)", w, h);
    AV_REQUIRE(ctx->d_simtab, AV_ESTATE, "av_simdet_generate: context has no draw table");
    const long long total = (long long)n_streams * n_frames;
    const int grid = (int)((total + 63) / 64);
    const SimRow* tab = (const SimRow*)ctx->d_simtab;
    if (n_frames == 1) {
        hipLaunchKernelGGL(simdet_kernel<true>, dim3(grid), dim3(64), 0, as_stream(stream), n_streams, n_frames, h, w, dcap,
                           frame_count, tab, ctx->d_cdf, det_n, det_box, det_cls, det_conf, status);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    hipLaunchKernelGGL(simdet_kernel<false>, dim3(grid), dim3(64), 0, as_stream(stream), n_streams, n_frames, h, w, dcap,
                       frame_count, tab, ctx->d_cdf, det_n, det_box, det_cls, det_conf, status);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(advance_counter_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, as_stream(stream), n_streams,
                       n_frames, frame_count);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

#endif  // AVHOT_DEVICE_ONLY
