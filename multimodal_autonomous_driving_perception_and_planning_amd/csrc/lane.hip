// L1-L7: lane detector pixel path.
//
// Reference: LaneDetector.detect (src/perception/lane_detector.py:178-218) and the OpenCV calls it
// makes (:63,69,72,79,83,89,94-101); semantics restated in oracle/c/lane_ref.c.
//
// Pipeline per frame (S frames per launch, u8 throughout, all integer arithmetic):
//   gray_blur_hist   BGR -> gray -> 5x5 binomial blur (LDS tiles, reflect-101), 256-bin histogram
//   thresholds       median of the blurred image from the histogram -> Canny lo/hi          (:79-81)
//   sobel_nms        3x3 Sobel, |gx|+|gy|, sector non-maximum suppression -> map {0 weak,1 none,2 strong}
//                    and union-find seeds for the candidates
//   ccl_merge        8-connected union-find over candidates; strong pixels carry the smaller labels, so a
//                    component's root is strong iff the component holds a strong pixel (exact hysteresis,
//                    independent of scheduling, no iteration-until-stable)
//   finalize         edge = candidate with a strong root; AND with the trapezoid ROI; per-row counts
//   compact          row-major list of ROI edge points (the order cv::HoughLinesP collects them in)
//   houghp           progressive probabilistic Hough, one workgroup per frame: lane = theta
//   fit              slope split, np.polyfit-equivalent quadratic, EMA with the previous fit, 50 points
#include "common.h"

#include <cmath>
#include <type_traits>

namespace {

constexpr int TW = 128, TH = 32;            // output tile of the pixel kernels
constexpr int NUMANGLE = 180;

struct LaneWs {                              // offsets (bytes) into the caller's workspace, per launch
    size_t blur, map, labels, edges, masked, hist, thr, rowcnt, nz, npts, accum, segs, nseg, tedge, rbits, kbits, total;
    int numrho;
};

__host__ inline size_t al256(size_t v) { return (v + 255) & ~size_t(255); }

__host__ LaneWs lane_layout(int S, int h, int w, int max_segments) {
    LaneWs L{};
    const size_t px = (size_t)S * h * w;
    L.numrho = (w + h) * 2 + 1;
    size_t o = 0;
    L.blur = o, o = al256(o + px);
    L.map = o, o = al256(o + px);
    L.labels = o, o = al256(o + px * 4);
    L.edges = o, o = al256(o + px);
    L.masked = o, o = al256(o + px);
    L.hist = o, o = al256(o + (size_t)S * 256 * 4);
    L.thr = o, o = al256(o + (size_t)S * 4 * 8);             // lo, hi (as doubles) , median, spare
    L.rowcnt = o, o = al256(o + (size_t)S * h * 4);
    L.nz = o, o = al256(o + px * 4);                          // packed x | y << 16
    L.npts = o, o = al256(o + (size_t)S * 4);
    L.accum = o, o = al256(o + (size_t)S * NUMANGLE * L.numrho * 4);
    L.segs = o, o = al256(o + (size_t)S * max_segments * 4 * 4);
    L.nseg = o, o = al256(o + (size_t)S * 4);
    // per 16 x 256 hysteresis tile: candidate bits of its top and bottom rows (8 + 8 words) and of its first and last columns
    // (16 + 16 bits), written by the tile pass for the border pass (TE_WORDS words per tile)
    L.tedge = o, o = al256(o + (size_t)S * ((h + 15) / 16) * ((w + 255) / 256) * 20 * 4);
    // one bit per pixel of the ROI's chunk box: candidates inside the ROI (tile pass -> resolve pass), kept edges (-> compaction);
    // sized for a box as large as the frame, rows padded to whole 32-bit words
    L.rbits = o, o = al256(o + (size_t)S * h * (((w + 15) / 16 + 2) / 2) * 4);
    L.kbits = o, o = al256(o + (size_t)S * h * (((w + 15) / 16 + 2) / 2) * 4);
    L.total = o;
    return L;
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
// one reflection, branch-free: p in [-(n-1), 2n-2], n >= 2 (the streaming kernels: rows -2 .. h+1 of frames with h >= 8)
__device__ __forceinline__ int reflect101_once(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
__device__ __forceinline__ int clampi(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); }

// ---- L1: gray + blur + histogram -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) gray_blur_hist_kernel(const uint8_t* __restrict__ bgr, int h, int w,
                                                             uint8_t* __restrict__ blur, unsigned* __restrict__ hist) {
    __shared__ uint8_t g[TH + 4][TW + 4 + 4];           // gray with halo 2
    __shared__ unsigned short t[TH + 4][TW];            // horizontal pass
    __shared__ unsigned lh[256];
    const int s = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const uint8_t* img = bgr + (size_t)s * h * w * 3;
    lh[tid] = 0;
    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
        const int r = i / (TW + 4), c = i - r * (TW + 4);
        const int yy = reflect101(y0 + r - 2, h), xx = reflect101(x0 + c - 2, w);
        const uint8_t* p = img + ((size_t)yy * w + xx) * 3;
        g[r][c] = (uint8_t)((1868 * p[0] + 9617 * p[1] + 4899 * p[2] + 8192) >> 14);
    }
    __syncthreads();
    for (int i = tid; i < (TH + 4) * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        t[r][c] = (unsigned short)(g[r][c] + 4 * g[r][c + 1] + 6 * g[r][c + 2] + 4 * g[r][c + 3] + g[r][c + 4]);
    }
    __syncthreads();
    // each thread: 16 consecutive output pixels of one row -> one 16-byte store
    const int r = tid >> 3, c0 = (tid & 7) * 16;
    const int y = y0 + r;
    if (y < h) {
        uint8_t o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + k;
            const int v = (t[r][c] + 4 * t[r + 1][c] + 6 * t[r + 2][c] + 4 * t[r + 3][c] + t[r + 4][c] + 128) >> 8;
            o[k] = (uint8_t)v;
            if (x0 + c < w) atomicAdd(&lh[v], 1u);
        }
        uint8_t* dst = blur + ((size_t)s * h + y) * w + x0 + c0;
        if (x0 + c0 + 16 <= w && (((size_t)dst) & 15) == 0) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
        } else {
            for (int k = 0; k < 16; ++k)
                if (x0 + c0 + k < w) dst[k] = o[k];
        }
    }
    __syncthreads();
    if (lh[tid]) atomicAdd(&hist[(size_t)s * 256 + tid], lh[tid]);
}

// ---- L2a: median -> thresholds; also clears the per-frame counters ----------------------------------------
__global__ void __launch_bounds__(256) thresholds_kernel(int h, int w, unsigned* __restrict__ hist,
                                                         double* __restrict__ thr, int* __restrict__ rowcnt,
                                                         int* __restrict__ npts, int* __restrict__ nseg) {
    __shared__ unsigned hh[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    hh[tid] = hist[(size_t)s * 256 + tid];
    hist[(size_t)s * 256 + tid] = 0;                    // ready for the next frame
    for (int i = tid; i < h; i += 256) rowcnt[(size_t)s * h + i] = 0;
    __syncthreads();
    // the two middle order statistics from the cumulative histogram: the first wave, four bins per lane, prefix over the lanes
    // (one thread walking the 256 bins was ~2 us of this 8-us launch); h * w < 2^31 (av_lane_detect), so the counts fit 32 bits
    if (tid < 64) {
        const unsigned a0 = hh[4 * tid], a1 = hh[4 * tid + 1], a2 = hh[4 * tid + 2], a3 = hh[4 * tid + 3], loc = a0 + a1 + a2 + a3;
        unsigned inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(inc, d, 64);
            if (tid >= d) inc += t;
        }
        const unsigned exc = inc - loc;
        const unsigned n = (unsigned)((long long)h * w), k1 = (n - 1) / 2, k2 = n / 2;
        auto bin_over = [&](unsigned k) -> int {          // first bin whose cumulative count exceeds k
            const bool here = inc > k && exc <= k;
            const int v = 4 * tid + (exc + a0 > k ? 0 : (exc + a0 + a1 > k ? 1 : (exc + a0 + a1 + a2 > k ? 2 : 3)));
            const unsigned long long m = __ballot(here);
            return __builtin_amdgcn_readlane(v, m ? __ffsll((long long)m) - 1 : 0);
        };
        const int v1 = bin_over(k1), v2 = bin_over(k2);
        if (tid != 0) return;
        const double med = ((double)v1 + (double)v2) / 2.0;          // np.median, even count
        const double l = 0.7 * med, u = 1.3 * med;
        double* o = thr + (size_t)s * 4;
        o[0] = (double)(int)(l > 0.0 ? l : 0.0);                      // int(max(0, 0.7*median))   :80
        o[1] = (double)(int)(u < 255.0 ? u : 255.0);                  // int(min(255, 1.3*median)) :81
        o[2] = med;
        npts[s] = 0, nseg[s] = 0;
    }
}

// ---- L2b: Sobel + NMS -> map, union-find seeds ----------------------------------------------------------------
__global__ void __launch_bounds__(256) sobel_nms_kernel(const uint8_t* __restrict__ blur, int h, int w,
                                                        const double* __restrict__ thr, uint8_t* __restrict__ map,
                                                        unsigned* __restrict__ labels) {
    __shared__ uint8_t b[TH + 4][TW + 4 + 4];
    __shared__ short gx[TH + 2][TW + 2], gy[TH + 2][TW + 2];
    __shared__ unsigned short mg[TH + 2][TW + 2];
    const int s = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const uint8_t* img = blur + (size_t)s * h * w;
    int lo = (int)thr[(size_t)s * 4], hi = (int)thr[(size_t)s * 4 + 1];
    if (lo > hi) { const int q = lo; lo = hi; hi = q; }
    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
        const int r = i / (TW + 4), c = i - r * (TW + 4);
        b[r][c] = img[(size_t)clampi(y0 + r - 2, h) * w + clampi(x0 + c - 2, w)];      // BORDER_REPLICATE
    }
    __syncthreads();
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
        const int r = i / (TW + 2), c = i - r * (TW + 2);         // image position (y0 + r - 1, x0 + c - 1)
        const int yy = y0 + r - 1, xx = x0 + c - 1;
        const int a00 = b[r][c], a01 = b[r][c + 1], a02 = b[r][c + 2];
        const int a10 = b[r + 1][c], a12 = b[r + 1][c + 2];
        const int a20 = b[r + 2][c], a21 = b[r + 2][c + 1], a22 = b[r + 2][c + 2];
        const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
        const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
        const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
        gx[r][c] = (short)dx, gy[r][c] = (short)dy;
        mg[r][c] = inside ? (unsigned short)(abs(dx) + abs(dy)) : 0;   // magnitude is 0 outside the image
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 16;
    const int y = y0 + r;
    if (y >= h) return;
    uint8_t o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + k, rr = r + 1, cc = c + 1;
        const int m = mg[rr][cc];
        int v = 1;
        if (m > lo) {
            const int xs = gx[rr][cc], ys = gy[rr][cc];
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;
            bool is_max;
            if (ay < tg22x) is_max = m > mg[rr][cc - 1] && m >= mg[rr][cc + 1];
            else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) is_max = m > mg[rr - 1][cc] && m >= mg[rr + 1][cc];
                else {
                    const int sg = (xs ^ ys) < 0 ? -1 : 1;
                    is_max = m > mg[rr - 1][cc - sg] && m > mg[rr + 1][cc + sg];
                }
            }
            if (is_max) v = m > hi ? 2 : 0;
        }
        o[k] = (uint8_t)v;
        const int x = x0 + c;
        if (v != 1 && x < w) {
            const unsigned idx = (unsigned)(y * w + x);
            labels[(size_t)s * h * w + idx] = v == 2 ? idx : (idx | 0x80000000u);   // strong labels sort first
        }
    }
    uint8_t* dst = map + ((size_t)s * h + y) * w + x0 + c0;
    if (x0 + c0 + 16 <= w && (((size_t)dst) & 15) == 0) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
    else
        for (int k = 0; k < 16; ++k)
            if (x0 + c0 + k < w) dst[k] = o[k];
}

// ---- L2c: hysteresis as connected components -------------------------------------------------------------------
__device__ __forceinline__ unsigned uf_find(unsigned* lab, unsigned v) {      // v, result: label values
    unsigned i = v & 0x7fffffffu;
    unsigned cur = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((cur & 0x7fffffffu) != i) {
        const unsigned p = cur & 0x7fffffffu;
        const unsigned nxt = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // path halving: labels only ever decrease along a path, so pointing i at its grandparent keeps the forest valid
        if (nxt != cur) atomicMin(&lab[i], nxt);
        i = p, cur = nxt;
    }
    return cur;
}
__device__ __forceinline__ void uf_union(unsigned* lab, unsigned a, unsigned b) {
    for (;;) {
        unsigned ra = uf_find(lab, a), rb = uf_find(lab, b);
        if (ra == rb) return;
        if (ra < rb) { const unsigned q = ra; ra = rb; rb = q; }             // ra: larger label value
        const unsigned old = atomicMin(&lab[ra & 0x7fffffffu], rb);
        if (old == ra) return;
        a = old, b = rb;
    }
}

__global__ void __launch_bounds__(256) ccl_merge_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                        unsigned* __restrict__ labels) {
    const int s = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t* m = map + (size_t)s * h * w;
    unsigned* lab = labels + (size_t)s * h * w;
    if (m[(size_t)y * w + x] == 1) return;
    const unsigned me = (unsigned)(y * w + x);
    if (x > 0 && m[(size_t)y * w + x - 1] != 1) uf_union(lab, me, me - 1);
    if (y > 0) {
        const size_t up = (size_t)(y - 1) * w;
        if (x > 0 && m[up + x - 1] != 1) uf_union(lab, me, (unsigned)(up + x - 1));
        if (m[up + x] != 1) uf_union(lab, me, (unsigned)(up + x));
        if (x + 1 < w && m[up + x + 1] != 1) uf_union(lab, me, (unsigned)(up + x + 1));
    }
}

struct Roi {
    int x0, x1, x2, x3, yt;
};
__device__ __forceinline__ void roi_bounds(const Roi& r, int h, int y, const int* rows, int& xl, int& xr) {
    if (rows) { xl = rows[2 * y], xr = rows[2 * y + 1]; return; }
    if (y < r.yt || y >= h) { xl = 1, xr = 0; return; }
    const long long den = h - r.yt, t = h - y;
    xl = (int)((2 * (r.x0 * den + (long long)(r.x1 - r.x0) * t) + den) / (2 * den));
    xr = (int)((2 * (r.x3 * den + (long long)(r.x2 - r.x3) * t) + den) / (2 * den));
}

__global__ void __launch_bounds__(256) finalize_kernel(const uint8_t* __restrict__ map, int h, int w,
                                                       unsigned* __restrict__ labels, Roi roi,
                                                       const int* __restrict__ roi_rows, uint8_t* __restrict__ edges,
                                                       uint8_t* __restrict__ masked, int* __restrict__ rowcnt) {
    const int s = blockIdx.z, y = blockIdx.y, tid = threadIdx.x;
    const uint8_t* m = map + ((size_t)s * h + y) * w;
    unsigned* lab = labels + (size_t)s * h * w;
    int xl, xr;
    roi_bounds(roi, h, y, roi_rows, xl, xr);
    int cnt = 0;
    for (int x = blockIdx.x * 1024 + tid * 4; x < w && x < (int)(blockIdx.x + 1) * 1024; x += 1024) {
        uint8_t e[4] = {0, 0, 0, 0}, k[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = x + q;
            if (xx < w && m[xx] != 1) {
                const unsigned root = uf_find(lab, (unsigned)(y * w + xx));
                if (!(root >> 31)) {
                    e[q] = 255;
                    if (xx >= xl && xx <= xr) k[q] = 255, ++cnt;
                }
            }
        }
        const size_t off = ((size_t)s * h + y) * w + x;
        if (x + 4 <= w && ((off & 3) == 0)) {
            if (edges) *reinterpret_cast<uchar4*>(edges + off) = make_uchar4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<uchar4*>(masked + off) = make_uchar4(k[0], k[1], k[2], k[3]);
        } else {
            for (int q = 0; q < 4 && x + q < w; ++q) {
                if (edges) edges[off + q] = e[q];
                masked[off + q] = k[q];
            }
        }
    }
    cnt = wave_sum_i(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&rowcnt[(size_t)s * h + y], cnt);
}

// ===== fast paths (w % 16 == 0, 16-byte aligned frames): every global access is a 16-byte vector ===========

// 16 sub-histograms (lane & 15) cut same-bin LDS-atomic conflicts from 64-way to 4-way on flat image areas.
__device__ __forceinline__ void hist_add(unsigned* lh16, int lane, int v, unsigned n) {
    atomicAdd(&lh16[(lane & 15) * 256 + v], n);
}

__global__ void __launch_bounds__(256) gray_blur_hist_fast(const uint8_t* __restrict__ bgr, int h, int w,
                                                           uint8_t* __restrict__ blur, unsigned* __restrict__ hist) {
    constexpr int RAWB = 432;                               // >= (TW + 4) * 3 + 15, multiple of 16
    __shared__ __attribute__((aligned(16))) uint8_t raw[TH + 4][RAWB];
    __shared__ uint8_t g[TH + 4][TW + 8];
    __shared__ unsigned short t[TH + 4][TW];
    __shared__ unsigned lh[16 * 256];
    const int s = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x, lane = tid & 63;
    const uint8_t* img = bgr + (size_t)s * h * w * 3;
    for (int i = tid; i < 16 * 256; i += 256) lh[i] = 0;
    // in-image column range this tile needs, as a 16-byte aligned byte range of the row
    const int cxa = x0 - 2 < 0 ? 0 : x0 - 2, cxb = x0 + TW + 2 > w ? w : x0 + TW + 2;
    const int b0 = (cxa * 3) & ~15, b1 = (cxb * 3 + 15) & ~15;          // w*3 % 16 == 0 -> b1 <= row bytes
    const int nch = (b1 - b0) >> 4;
    for (int i = tid; i < (TH + 4) * nch; i += 256) {
        const int r = i / nch, c = i - r * nch;
        const int yy = reflect101(y0 + r - 2, h);
        *reinterpret_cast<uint4*>(&raw[r][c * 16]) =
            *reinterpret_cast<const uint4*>(img + (size_t)yy * w * 3 + b0 + c * 16);
    }
    __syncthreads();
    for (int i = tid; i < (TH + 4) * (TW + 4); i += 256) {
        const int r = i / (TW + 4), c = i - r * (TW + 4);
        const int xx = reflect101(x0 + c - 2, w);
        const uint8_t* p = &raw[r][xx * 3 - b0];
        g[r][c] = (uint8_t)((1868 * p[0] + 9617 * p[1] + 4899 * p[2] + 8192) >> 14);
    }
    __syncthreads();
    for (int i = tid; i < (TH + 4) * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        t[r][c] = (unsigned short)(g[r][c] + 4 * g[r][c + 1] + 6 * g[r][c + 2] + 4 * g[r][c + 3] + g[r][c + 4]);
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 16;
    const int y = y0 + r;
    if (y < h && x0 + c0 < w) {
        uint8_t o[16];
        int run_v = -1;
        unsigned run_n = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + k;
            const int v = (t[r][c] + 4 * t[r + 1][c] + 6 * t[r + 2][c] + 4 * t[r + 3][c] + t[r + 4][c] + 128) >> 8;
            o[k] = (uint8_t)v;
            if (v == run_v) ++run_n;
            else {
                if (run_n) hist_add(lh, lane, run_v, run_n);
                run_v = v, run_n = 1;
            }
        }
        hist_add(lh, lane, run_v, run_n);
        *reinterpret_cast<uint4*>(blur + ((size_t)s * h + y) * w + x0 + c0) = *reinterpret_cast<const uint4*>(o);
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += lh[q * 256 + tid];
    if (tot) atomicAdd(&hist[(size_t)s * 256 + tid], tot);
}

__global__ void __launch_bounds__(256) sobel_nms_fast(const uint8_t* __restrict__ blur, int h, int w,
                                                      const double* __restrict__ thr, uint8_t* __restrict__ map,
                                                      unsigned* __restrict__ labels) {
    constexpr int RB = TW + 32;                              // tile row bytes incl. 16-byte aligned halo
    __shared__ __attribute__((aligned(16))) uint8_t b[TH + 4][RB];
    __shared__ short gx[TH + 2][TW + 2], gy[TH + 2][TW + 2];
    __shared__ unsigned short mg[TH + 2][TW + 2];
    const int s = blockIdx.z, x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, tid = threadIdx.x;
    const uint8_t* img = blur + (size_t)s * h * w;
    int lo = (int)thr[(size_t)s * 4], hi = (int)thr[(size_t)s * 4 + 1];
    if (lo > hi) { const int q = lo; lo = hi; hi = q; }
    // b[r][16 + c] holds image column x0 + c; chunks x0-16 .. x0+TW+15 (clamped chunks duplicate the border)
    constexpr int NCH = RB / 16;
    for (int i = tid; i < (TH + 4) * NCH; i += 256) {
        const int r = i / NCH, c = i - r * NCH;
        const int yy = clampi(y0 + r - 2, h);
        int xs = x0 - 16 + c * 16;
        xs = xs < 0 ? 0 : (xs > w - 16 ? w - 16 : xs);
        *reinterpret_cast<uint4*>(&b[r][c * 16]) = *reinterpret_cast<const uint4*>(img + (size_t)yy * w + xs);
    }
    __syncthreads();
    auto px = [&](int r, int xx) -> int {                     // BORDER_REPLICATE column lookup inside the tile
        const int xc = clampi(xx, w);
        int off = xc - x0 + 16;
        if (x0 == 0 && off < 16) off = xc + 16;               // left chunk was clamped to columns 0..15
        if (x0 + TW >= w && off >= 16 + TW) off = xc - (w - 16) + 16 + TW;   // right chunk clamped to w-16..w-1
        return b[r][off];
    };
    for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
        const int r = i / (TW + 2), c = i - r * (TW + 2);
        const int yy = y0 + r - 1, xx = x0 + c - 1;
        const int a00 = px(r, xx - 1), a01 = px(r, xx), a02 = px(r, xx + 1);
        const int a10 = px(r + 1, xx - 1), a12 = px(r + 1, xx + 1);
        const int a20 = px(r + 2, xx - 1), a21 = px(r + 2, xx), a22 = px(r + 2, xx + 1);
        const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
        const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
        const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
        gx[r][c] = (short)dx, gy[r][c] = (short)dy;
        mg[r][c] = inside ? (unsigned short)(abs(dx) + abs(dy)) : 0;
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 16;
    const int y = y0 + r;
    if (y >= h || x0 + c0 >= w) return;
    uint8_t o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + k, rr = r + 1, cc = c + 1;
        const int m = mg[rr][cc];
        int v = 1;
        if (m > lo) {
            const int xs = gx[rr][cc], ys = gy[rr][cc];
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;
            bool is_max;
            if (ay < tg22x) is_max = m > mg[rr][cc - 1] && m >= mg[rr][cc + 1];
            else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) is_max = m > mg[rr - 1][cc] && m >= mg[rr + 1][cc];
                else {
                    const int sg = (xs ^ ys) < 0 ? -1 : 1;
                    is_max = m > mg[rr - 1][cc - sg] && m > mg[rr + 1][cc + sg];
                }
            }
            if (is_max) v = m > hi ? 2 : 0;
        }
        o[k] = (uint8_t)v;
        if (v != 1) {
            const unsigned idx = (unsigned)(y * w + x0 + c);
            labels[(size_t)s * h * w + idx] = v == 2 ? idx : (idx | 0x80000000u);
        }
    }
    *reinterpret_cast<uint4*>(map + ((size_t)s * h + y) * w + x0 + c0) = *reinterpret_cast<const uint4*>(o);
}

// ===== streaming kernels (w % 4 == 0): no LDS tiles; a lane owns 4 adjacent pixels of a column strip and
// slides down the rows, exchanging row neighbours with DPP.  A wave covers 64 chunks = 256 columns of which
// lanes 0 and 63 are halo (strip pitch 248 columns); image borders are resolved by permuting the edge lane's
// own pixels (reflect-101 for the blur, replicate for Sobel).

__device__ __forceinline__ unsigned dpp_prev_u32(unsigned v) {      // value of lane-1 (0 for lane 0)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned dpp_next_u32(unsigned v) {      // value of lane+1 (0 for lane 63)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

constexpr int SW = 248;                 // output columns per wave strip (62 lanes x 4)
constexpr int SROWS = 48;               // output rows per wave
constexpr int SPF = 3;                  // rows fetched ahead
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) gray_blur_hist_stream(const uint8_t* __restrict__ bgr, int h, int w,
                                                             uint8_t* __restrict__ blur, unsigned* __restrict__ hist) {
    __shared__ unsigned lh[16 * 256];
    const int s = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 16 * 256; i += 256) lh[i] = 0;
    __syncthreads();
    // the four waves of a workgroup take neighbouring strips of the same band of rows (strip fastest over the grid):
    // together they read a contiguous piece of every row
    const int nstrips = (w + SW - 1) / SW, wv = blockIdx.x * 4 + wid;
    const int strip = wv % nstrips, x0 = strip * SW - 4;        // column of lane 0's first pixel (halo lane)
    const int yb = (wv / nstrips) * SROWS;                      // first output row of this wave
    const int x = x0 + 4 * lane;                                // this lane's columns x .. x+3
    const bool xin = x >= 0 && x + 4 <= w;                      // chunk fully inside the image (w % 4 == 0)
    const bool out_lane = lane >= 1 && lane <= 62 && xin;
    const uint8_t* img = bgr + (size_t)s * h * w * 3;
    uint8_t* dst = blur + (size_t)s * h * w;
    if (yb < h) {
        // horizontal pass results of the last five rows, two u16 pairs per row (pixels 0,1 | 2,3)
        unsigned ra[5], rb[5];
        const int y_end = (yb + SROWS < h ? yb + SROWS : h);
        // A wave consumes one 768-byte row segment per trip and nothing else hides the load behind it: keep the next
        // SPF rows in flight.  Unconditional loads (halo lanes and rows past the end re-read a valid address), so the
        // compiler can count them and wait for exactly the oldest one.
        unsigned fa[SPF], fb[SPF], fc[SPF];
        const uint8_t* col = img + (size_t)(xin ? x : 0) * 3;
        const unsigned pitch = (unsigned)w * 3u;
        auto fetch = [&](int yy, unsigned& a, unsigned& b, unsigned& c) {
            const int ys = reflect101_once(yy < y_end + 1 ? yy : y_end + 1, h);
            const unsigned* p = reinterpret_cast<const unsigned*>(col + (size_t)ys * pitch);
            a = p[0], b = p[1], c = p[2];
        };
#pragma unroll
        for (int q = 0; q < SPF; ++q) fetch(yb - 2 + q, fa[q], fb[q], fc[q]);
        for (int yy = yb - 2; yy < y_end + 2; ++yy) {
            unsigned g = 0;                                     // 4 gray bytes, pixel 0 in the low byte
            const unsigned a = fa[0], b = fb[0], c = fc[0];     // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
#pragma unroll
            for (int q = 0; q + 1 < SPF; ++q) fa[q] = fa[q + 1], fb[q] = fb[q + 1], fc[q] = fc[q + 1];
            fetch(yy + SPF, fa[SPF - 1], fb[SPF - 1], fc[SPF - 1]);
            if (xin) {
                // gray = (1868 B + 9617 G + 4899 R + 8192) >> 14: B and G spread to a u16 pair by one byte permute, one
                // dot2 against (1868, 9617), one 24-bit multiply-add for R (all full rate)
                const u16x2_t wbg = {1868, 9617};
                auto gray = [&](unsigned bg_pair, unsigned rr) {
                    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, bg_pair), wbg, (unsigned)__umul24(4899u, rr) + 8192u, false) >> 14;
                };
                // v_perm_b32 selectors: bytes 0-3 = second operand, 4-7 = first operand, 0x0C = zero
                const unsigned g0 = gray(__builtin_amdgcn_perm(0u, a, 0x0C010C00u), (a >> 16) & 255u);          // B0 G0 | R0
                const unsigned g1 = gray(__builtin_amdgcn_perm(b, a, 0x0C040C03u), (b >> 8) & 255u);           // B1 (a.3) G1 (b.0) | R1
                const unsigned g2 = gray(__builtin_amdgcn_perm(0u, b, 0x0C030C02u), c & 255u);                  // B2 G2 | R2
                const unsigned g3 = gray(__builtin_amdgcn_perm(0u, c, 0x0C020C01u), c >> 24);                   // B3 G3 | R3
                g = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
            }
            unsigned gp = dpp_prev_u32(g), gn = dpp_next_u32(g);
            // reflect-101 at the image borders: columns -2,-1 are columns 2,1; columns w,w+1 are w-2,w-3
            if (x == 0) gp = ((g >> 16) & 255u) << 16 | ((g >> 8) & 255u) << 24;          // [.,.,g2,g1]
            if (x + 4 == w) gn = ((g >> 16) & 255u) | (((g >> 8) & 255u) << 8);            // [g2,g1,.,.]
            const unsigned m2 = gp >> 16 & 255u, m1 = gp >> 24, p0 = g & 255u, p1 = (g >> 8) & 255u, p2 = (g >> 16) & 255u,
                           p3 = g >> 24, n0 = gn & 255u, n1 = (gn >> 8) & 255u;
            const unsigned h0 = m2 + 4 * m1 + 6 * p0 + 4 * p1 + p2, h1 = m1 + 4 * p0 + 6 * p1 + 4 * p2 + p3;
            const unsigned h2 = p0 + 4 * p1 + 6 * p2 + 4 * p3 + n0, h3 = p1 + 4 * p2 + 6 * p3 + 4 * n0 + n1;
#pragma unroll
            for (int q = 0; q < 4; ++q) ra[q] = ra[q + 1], rb[q] = rb[q + 1];
            ra[4] = h0 | (h1 << 16), rb[4] = h2 | (h3 << 16);
            const int yo = yy - 2;                              // output row completed by this input row
            if (yo >= yb && out_lane) {
                // vertical pass on packed u16 pairs: max 16*4080 + 128 < 65536, no carry between the halves
                const unsigned va = ra[0] + ra[4] + 4u * (ra[1] + ra[3]) + 6u * ra[2] + 0x00800080u;
                const unsigned vb = rb[0] + rb[4] + 4u * (rb[1] + rb[3]) + 6u * rb[2] + 0x00800080u;
                const unsigned o0 = (va >> 8) & 255u, o1 = va >> 24, o2 = (vb >> 8) & 255u, o3 = vb >> 24;
                *reinterpret_cast<unsigned*>(dst + (size_t)yo * w + x) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
                // histogram with run merging
                unsigned* hl = lh + (lane & 15) * 256;
                if (o0 == o1 && o1 == o2 && o2 == o3) atomicAdd(&hl[o0], 4u);
                else {
                    atomicAdd(&hl[o0], 1u), atomicAdd(&hl[o1], 1u), atomicAdd(&hl[o2], 1u), atomicAdd(&hl[o3], 1u);
                }
            }
        }
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += lh[q * 256 + tid];
    if (tot) atomicAdd(&hist[(size_t)s * 256 + tid], tot);
}

// packed 16-bit pairs in a 32-bit register (v_pk_*_i16 / _u16)
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2_t, a) - __builtin_bit_cast(s16x2_t, b));
}
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2_t, a) + __builtin_bit_cast(s16x2_t, b));
}
__device__ __forceinline__ unsigned pk_abs16(unsigned a) {
    const s16x2_t v = __builtin_bit_cast(s16x2_t, a);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, -v));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

__global__ void __launch_bounds__(256) sobel_nms_stream(const uint8_t* __restrict__ blur, int h, int w,
                                                        const double* __restrict__ thr, uint8_t* __restrict__ map,
                                                        unsigned* __restrict__ labels) {
    const int s = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nstrips = (w + SW - 1) / SW, wv = blockIdx.x * 4 + wid;       // strip fastest, as in gray_blur_hist_stream
    const int strip = wv % nstrips, x0 = strip * SW - 4;
    const int yb = (wv / nstrips) * SROWS;
    if (yb >= h) return;
    const int x = x0 + 4 * lane;
    const bool xin = x >= 0 && x + 4 <= w;
    const bool out_lane = lane >= 1 && lane <= 62 && xin;
    const uint8_t* img = blur + (size_t)s * h * w;
    int lo = (int)thr[(size_t)s * 4], hi = (int)thr[(size_t)s * 4 + 1];
    if (lo > hi) { const int q = lo; lo = hi; hi = q; }
    const int y_end = (yb + SROWS < h ? yb + SROWS : h);
    // Everything per pixel is 16 bits wide (bytes, |gradients| <= 1020, magnitudes <= 2040), so the lane's four pixels
    // travel as two packed pairs -- X02 = (x0 | x2 << 16), X13 = (x1 | x3 << 16) -- plus XE = (column x-1 | column x+4
    // << 16) from the neighbouring lanes, and Sobel is packed 16-bit arithmetic: 40 instructions per row instead of 85.
    // Rings: blurred rows (3), magnitude rows (3), the middle row's gradients (2).
    unsigned p02[3] = {0, 0, 0}, p13[3] = {0, 0, 0}, pe[3] = {0, 0, 0};
    unsigned m02[3] = {0, 0, 0}, m13[3] = {0, 0, 0}, me[3] = {0, 0, 0};
    unsigned gx02[2] = {0, 0}, gx13[2] = {0, 0}, gy02[2] = {0, 0}, gy13[2] = {0, 0};
    unsigned fr[SPF];                               // the next SPF rows, in flight (see gray_blur_hist_stream)
    const uint8_t* col = img + (xin ? x : 0);
    auto fetch = [&](int yy) {
        const int ys = clampi(yy < y_end + 1 ? yy : y_end + 1, h);
        return *reinterpret_cast<const unsigned*>(col + (size_t)ys * (unsigned)w);
    };
#pragma unroll
    for (int q = 0; q < SPF; ++q) fr[q] = fetch(yb - 2 + q);
    for (int yy = yb - 2; yy < y_end + 2; ++yy) {
        // ---- stage 1: blurred row yy (replicate at the borders) -----------------------------------------
        const unsigned c = xin ? fr[0] : 0u;
#pragma unroll
        for (int q = 0; q + 1 < SPF; ++q) fr[q] = fr[q + 1];
        fr[SPF - 1] = fetch(yy + SPF);
        unsigned lft = dpp_prev_u32(c) >> 24, rgt = dpp_next_u32(c) & 255u;
        if (x == 0) lft = c & 255u;
        if (x + 4 == w) rgt = c >> 24;
        p02[0] = p02[1], p02[1] = p02[2], p02[2] = c & 0x00FF00FFu;
        p13[0] = p13[1], p13[1] = p13[2], p13[2] = (c >> 8) & 0x00FF00FFu;
        pe[0] = pe[1], pe[1] = pe[2], pe[2] = lft | (rgt << 16);
        // ---- stage 2: Sobel of row yy-1 from blurred rows yy-2, yy-1, yy -----------------------------------
        const int ym = yy - 1;
        m02[0] = m02[1], m02[1] = m02[2], m13[0] = m13[1], m13[1] = m13[2], me[0] = me[1], me[1] = me[2];
        gx02[0] = gx02[1], gx13[0] = gx13[1], gy02[0] = gy02[1], gy13[0] = gy13[1];
        {
            // columns x-1 .. x+4: vertical smooth V = r0 + 2 r1 + r2 (<= 1020, plain adds) and difference D = r2 - r0
            const unsigned v02 = p02[0] + 2u * p02[1] + p02[2], v13 = p13[0] + 2u * p13[1] + p13[2], ve = pe[0] + 2u * pe[1] + pe[2];
            const unsigned d02 = pk_sub16(p02[2], p02[0]), d13 = pk_sub16(p13[2], p13[0]), de = pk_sub16(pe[2], pe[0]);
            // dx_k = V[k+2] - V[k]:  (dx0, dx2) = V13 - (VE.lo, V13.lo);  (dx1, dx3) = (V02.hi, VE.hi) - V02
            const unsigned dx02 = pk_sub16(v13, (ve & 0xFFFFu) | (v13 << 16));
            const unsigned dx13 = pk_sub16((v02 >> 16) | (ve & 0xFFFF0000u), v02);
            // dy_k = D[k] + 2 D[k+1] + D[k+2]:  (dy0, dy2) = (DE.lo, D13.lo) + 2 D02 + D13;  (dy1, dy3) = D02 + 2 D13 + (D02.hi, DE.hi)
            const unsigned dy02 = pk_add16(pk_add16((de & 0xFFFFu) | (d13 << 16), pk_add16(d02, d02)), d13);
            const unsigned dy13 = pk_add16(pk_add16(d02, pk_add16(d13, d13)), (d02 >> 16) | (de & 0xFFFF0000u));
            gx02[1] = dx02, gx13[1] = dx13, gy02[1] = dy02, gy13[1] = dy13;
            const bool in = ym >= 0 && ym < h && xin;                    // magnitude is 0 outside the image
            m02[2] = in ? pk_abs16(dx02) + pk_abs16(dy02) : 0u;           // <= 2040 per half: plain add
            m13[2] = in ? pk_abs16(dx13) + pk_abs16(dy13) : 0u;
            // column x-1 = the previous lane's pixel 3, column x+4 = the next lane's pixel 0 (0 outside: halo lanes hold 0)
            me[2] = (dpp_prev_u32(m13[2]) >> 16) | (dpp_next_u32(m02[2]) << 16);
        }
        // ---- stage 3: NMS of row yy-2 from magnitude rows yy-3, yy-2, yy-1 ---------------------------------
        const int yo = yy - 2;
        if (yo >= yb && yo < y_end && out_lane) {
            unsigned o = 0x01010101u;
            const unsigned mx = pk_max_u16(m02[1], m13[1]);
            const int mmax = (int)max(mx & 0xFFFFu, mx >> 16);
            if (mmax > lo) {
                int mm[3][6];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    mm[q][0] = (int)(me[q] & 0xFFFFu), mm[q][5] = (int)(me[q] >> 16);
                    mm[q][1] = (int)(m02[q] & 0xFFFFu), mm[q][3] = (int)(m02[q] >> 16);
                    mm[q][2] = (int)(m13[q] & 0xFFFFu), mm[q][4] = (int)(m13[q] >> 16);
                }
                const int gxs[4] = {(int)(short)(gx02[0] & 0xFFFFu), (int)(short)(gx13[0] & 0xFFFFu), (int)(short)(gx02[0] >> 16),
                                    (int)(short)(gx13[0] >> 16)};
                const int gys[4] = {(int)(short)(gy02[0] & 0xFFFFu), (int)(short)(gy13[0] & 0xFFFFu), (int)(short)(gy02[0] >> 16),
                                    (int)(short)(gy13[0] >> 16)};
                o = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = mm[1][k + 1];
                    int v = 1;
                    if (m > lo) {
                        const int xs = gxs[k], ysg = gys[k];
                        const int ax = abs(xs), ay = abs(ysg) << 15;
                        const int tg22x = __mul24(ax, 13573);                    // |gx| <= 1020
                        bool is_max;
                        if (ay < tg22x) is_max = m > mm[1][k] && m >= mm[1][k + 2];
                        else {
                            const int tg67x = tg22x + (ax << 16);
                            if (ay > tg67x) is_max = m > mm[0][k + 1] && m >= mm[2][k + 1];
                            else {
                                const bool neg = (xs ^ ysg) < 0;
                                const int a1 = neg ? mm[0][k + 2] : mm[0][k], a2 = neg ? mm[2][k] : mm[2][k + 2];
                                is_max = m > a1 && m > a2;
                            }
                        }
                        if (is_max) v = m > hi ? 2 : 0;
                    }
                    o |= (unsigned)v << (8 * k);
                    if (v != 1) {
                        const unsigned idx = (unsigned)(yo * w + x + k);
                        labels[(size_t)s * h * w + idx] = v == 2 ? idx : (idx | 0x80000000u);
                    }
                }
            }
            *reinterpret_cast<unsigned*>(map + ((size_t)s * h + yo) * w + x) = o;
        }
    }
}


// ---- fused front end: BGR -> gray -> 5x5 blur (+ histogram) -> Sobel -> non-maximum suppression, ONE pass ---------
// Canny's thresholds come from the median of the whole blurred frame, which is why the reference's order forces the
// blurred image out to memory and back.  But the thresholds are not needed to decide WHETHER a pixel is a maximum
// along its gradient, only to classify the maxima afterwards: so this kernel streams a frame once, keeps the blurred
// rows in a register ring, and writes
//     nm[y][x] = 0                    if the pixel is not a local maximum along its gradient direction
//              = min(m / 2, 255)      if it is, m = |gx| + |gy|
// m is always EVEN (gx and gy are both congruent to the sum of the four corner pixels modulo 2), and the thresholds
// are at most 255, so "m > lo" == "m/2 > lo/2" and a value saturated at 255 compares like the full magnitude: the
// hysteresis pass classifies with  candidate = nm > (lo >> 1),  strong = nm > (hi >> 1)  -- exactly cv::Canny's sets.
// That removes one full-frame write and one read (the blurred image) and the second pass's reloads.
// Same streaming scheme as the two kernels it replaces (lane = 4 adjacent pixels of a 248-column strip, neighbours
// by DPP); rows: gray rows feed a 5-deep ring of horizontal sums -> blurred row b -> Sobel of row b-1 -> NMS of row
// b-2.  Border rules as in OpenCV: reflect-101 for the blur, replicate for Sobel's reads of the blurred image
// (blurred(-1) == blurred(0): the same register row is pushed again), magnitude 0 outside the image.
// output rows per wave: a template parameter (8 halo rows are recomputed per band; 45 gives 1280x720 frames 16 bands)

// packed 16-bit helpers (VOP3P): a register holds two pixels, X02 = (x0 | x2 << 16), X13 = (x1 | x3 << 16) -- with this
// interleaving the right neighbours of pixels (0,2) are exactly register X13 and the left neighbours of (1,3) exactly X02
__device__ __forceinline__ unsigned pk_addu(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b));
}
__device__ __forceinline__ unsigned pk_mulu(unsigned a, unsigned short k) {
    const u16x2_t kk = {k, k};
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, a) * kk);
}
__device__ __forceinline__ unsigned pk_mulv(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, a) * __builtin_bit_cast(u16x2_t, b));
}
__device__ __forceinline__ unsigned pk_shru(unsigned a, int n) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, a) >> (unsigned short)n);
}
__device__ __forceinline__ unsigned pk_shlu(unsigned a, int n) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, a) << (unsigned short)n);
}
__device__ __forceinline__ unsigned pk_sar15(unsigned a) {            // 0xFFFF where the half is negative, else 0
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2_t, a) >> (short)15);
}
__device__ __forceinline__ unsigned pk_subsat(unsigned a, unsigned b) {      // max(a - b, 0) per half
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ unsigned pk_minu(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
// 16-bit halves of two registers -> one pair (v_perm_b32: selector bytes 0-3 = second operand, 4-7 = first)
__device__ __forceinline__ unsigned lo_lo(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x05040100u); }   // x.lo | y.lo << 16
__device__ __forceinline__ unsigned hi_lo(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x05040302u); }   // x.hi | y.lo << 16
__device__ __forceinline__ unsigned lo_hi(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x07060100u); }   // x.lo | y.hi << 16
__device__ __forceinline__ unsigned hi_hi(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x07060302u); }   // x.hi | y.hi << 16
__device__ __forceinline__ unsigned bsel(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }     // v_bfi_b32
// hides where a mask came from: otherwise the compiler turns sign-mask + v_bfi back into per-half compares and selects
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <bool KEEP_BLUR, int FROWS>
__global__ void __launch_bounds__(256) front_stream(const uint8_t* __restrict__ bgr, int h, int w, uint8_t* __restrict__ blur,
                                                    uint8_t* __restrict__ nm, unsigned* __restrict__ hist) {
    __shared__ unsigned lh[16 * 256];
    const int s = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform: rows, bounds and row tests stay scalar
    for (int i = tid; i < 16 * 256; i += 256) lh[i] = 0;
    __syncthreads();
    const int nstrips = (w + SW - 1) / SW, wv = blockIdx.x * 4 + wid;
    const int strip = wv % nstrips, x0 = strip * SW - 4;
    const int yb = (wv / nstrips) * FROWS;
    const int x = x0 + 4 * lane;
    const bool xin = x >= 0 && x + 4 <= w;
    const bool out_lane = lane >= 1 && lane <= 62 && xin;
    const bool at_left = x == 0, at_right = x + 4 == w;
    const unsigned lane_mask = xin ? 0xFFFFFFFFu : 0u;
    const uint8_t* img = bgr + (size_t)s * h * w * 3;
    const size_t fo = (size_t)s * h * w;
    if (yb < h) {
        const int y_end = (yb + FROWS < h ? yb + FROWS : h);
        // Iteration r consumes gray row r (reflect-101) and completes: horizontal sums H[r], U[r-1] = H[r-2] + 2 H[r-1] + H[r],
        // blurred row b = r-2 = (U[r-3] + 2 U[r-2] + U[r-1] + 128) >> 8   ((1 2 1) * (1 2 1) = 1 4 6 4 1),
        // Sobel + direction class of row r-3, NMS of row r-4.  Every ring is three deep and the loop is unrolled by
        // three, so ring slots are compile-time registers and nothing is ever moved.
        const int r_first = yb - 4, r_last = y_end + 3;
        unsigned H02[3] = {0, 0, 0}, H13[3] = {0, 0, 0}, U02[3] = {0, 0, 0}, U13[3] = {0, 0, 0};
        unsigned P02[3] = {0, 0, 0}, P13[3] = {0, 0, 0}, PE[3] = {0, 0, 0};
        unsigned M02[3] = {0, 0, 0}, M13[3] = {0, 0, 0}, ME[3] = {0, 0, 0};
        unsigned HZ02[3] = {0, 0, 0}, HZ13[3] = {0, 0, 0}, VT02[3] = {0, 0, 0}, VT13[3] = {0, 0, 0}, NG02[3] = {0, 0, 0},
                 NG13[3] = {0, 0, 0};
        unsigned fa[3], fb[3], fc[3];                                    // the next three BGR rows, in flight
        const uint8_t* col = img + (size_t)(xin ? x : 0) * 3;
        const unsigned pitch = (unsigned)w * 3u;
        auto fetch = [&](int yy, unsigned& a, unsigned& b, unsigned& c) {
            const int ys = reflect101_once(yy < r_last ? yy : r_last, h);
            const unsigned* p = reinterpret_cast<const unsigned*>(col + (size_t)ys * pitch);
            a = p[0], b = p[1], c = p[2];
        };
#pragma unroll
        for (int q = 0; q < 3; ++q) fetch(r_first + q, fa[q], fb[q], fc[q]);
        unsigned* hl = lh + (lane & 15) * 256;
        for (int r0 = r_first; r0 <= r_last; r0 += 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = r0 + k;                                    // (up to two rows past r_last: nothing they produce is kept)
                const int k1 = (k + 2) % 3, k2 = (k + 1) % 3;            // slots of the previous and the one before
                // ---- gray row r: 4 pixels -> pairs G02, G13 -----------------------------------------------------
                const unsigned a = fa[k], b = fb[k], c = fc[k];          // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                fetch(r + 3, fa[k], fb[k], fc[k]);
                const u16x2_t wbg = {1868, 9617};
                auto gray = [&](unsigned bg_pair, unsigned rr) {
                    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, bg_pair), wbg, (unsigned)__umul24(4899u, rr) + 8192u, false) >> 14;
                };
                const unsigned g0 = gray(__builtin_amdgcn_perm(0u, a, 0x0C010C00u), (a >> 16) & 255u);
                const unsigned g1 = gray(__builtin_amdgcn_perm(b, a, 0x0C040C03u), (b >> 8) & 255u);
                const unsigned g2 = gray(__builtin_amdgcn_perm(0u, b, 0x0C030C02u), c & 255u);
                const unsigned g3 = gray(__builtin_amdgcn_perm(0u, c, 0x0C020C01u), c >> 24);
                const unsigned G02 = g0 | (g2 << 16), G13 = g1 | (g3 << 16);
                // ---- horizontal 1 4 6 4 1 with the neighbours' pixels (reflect-101 at the image's left / right edge) ----
                const unsigned R21 = hi_lo(G02, G13);                    // (g2, g1): columns -2,-1 and w,w+1 mirror to these
                unsigned gprev = dpp_prev_u32(hi_hi(G02, G13));          // the previous lane's (g2, g3) = columns x-2, x-1
                unsigned gnext = dpp_next_u32(lo_lo(G02, G13));          // the next lane's (g0, g1) = columns x+4, x+5
                gprev = at_left ? R21 : gprev, gnext = at_right ? R21 : gnext;
                const unsigned L2 = lo_lo(gprev, G02), L1 = hi_lo(gprev, G13);      // (g-2, g0), (g-1, g1)
                const unsigned R2 = hi_lo(G02, gnext), R3 = hi_hi(G13, gnext);      // (g2, g4), (g3, g5)
                H02[k] = pk_addu(pk_addu(L2, R2), pk_addu(pk_shlu(pk_addu(L1, G13), 2), pk_mulu(G02, 6)));
                H13[k] = pk_addu(pk_addu(L1, R3), pk_addu(pk_shlu(pk_addu(G02, R2), 2), pk_mulu(G13, 6)));
                // ---- vertical: U[r-1], then blurred row b = r-2 -------------------------------------------------
                U02[k] = pk_addu(pk_addu(H02[k2], H02[k]), pk_shlu(H02[k1], 1));
                U13[k] = pk_addu(pk_addu(H13[k2], H13[k]), pk_shlu(H13[k1], 1));
                const unsigned V02 = pk_addu(pk_addu(U02[k2], U02[k]), pk_addu(pk_shlu(U02[k1], 1), 0x00800080u));
                const unsigned V13 = pk_addu(pk_addu(U13[k2], U13[k]), pk_addu(pk_shlu(U13[k1], 1), 0x00800080u));
                const unsigned p02 = pk_shru(V02, 8), p13 = pk_shru(V13, 8);        // (o0, o2), (o1, o3)
                P02[k] = p02, P13[k] = p13;
                const int bl = r - 2;
                if (bl >= yb && bl < y_end) {                            // this wave owns the blurred row: histogram (+ debug copy)
                    if (out_lane) {
                        const unsigned o0 = p02 & 0xFFFFu, o2 = p02 >> 16, o1 = p13 & 0xFFFFu, o3 = p13 >> 16;
                        if (KEEP_BLUR) *reinterpret_cast<unsigned*>(blur + fo + (size_t)bl * w + x) = p02 | (p13 << 8);
                        if (p02 == p13 && o0 == o2) atomicAdd(&hl[o0], 4u);
                        else atomicAdd(&hl[o0], 1u), atomicAdd(&hl[o1], 1u), atomicAdd(&hl[o2], 1u), atomicAdd(&hl[o3], 1u);
                    }
                }
                {   // blurred neighbours across the lane border: PE = (column x-1 | column x+4 << 16), replicated at the image edge
                    unsigned dp = dpp_prev_u32(p13), dn = dpp_next_u32(p02);        // .hi = o3 of the previous lane, .lo = o0 of the next
                    dp = at_left ? (p02 << 16) : dp, dn = at_right ? (p13 >> 16) : dn;
                    PE[k] = hi_lo(dp, dn);
                }
                // ---- Sobel of row ym = r-3 from blurred rows r-4, r-3, r-2 (replicated at the image's top / bottom) ----
                const int ym = r - 3;
                {
                    unsigned t02 = P02[k2], t13 = P13[k2], te = PE[k2], b02 = P02[k], b13 = P13[k], be = PE[k];
                    const unsigned c02 = P02[k1], c13 = P13[k1], ce = PE[k1];
                    if (ym == 0) t02 = c02, t13 = c13, te = ce;
                    if (ym == h - 1) b02 = c02, b13 = c13, be = ce;
                    const unsigned v02 = pk_addu(pk_addu(t02, b02), pk_shlu(c02, 1)), v13 = pk_addu(pk_addu(t13, b13), pk_shlu(c13, 1)),
                                   ve = pk_addu(pk_addu(te, be), pk_shlu(ce, 1));
                    const unsigned d02 = pk_sub16(b02, t02), d13 = pk_sub16(b13, t13), de = pk_sub16(be, te);
                    const unsigned dx02 = pk_sub16(v13, lo_lo(ve, v13));                       // V[k+1] - V[k-1] for pixels 0, 2
                    const unsigned dx13 = pk_sub16(hi_hi(v02, ve), v02);                       //                  for pixels 1, 3
                    const unsigned dy02 = pk_add16(pk_add16(lo_lo(de, d13), pk_add16(d02, d02)), d13);
                    const unsigned dy13 = pk_add16(pk_add16(d02, pk_add16(d13, d13)), hi_hi(d02, de));
                    const unsigned ax02 = pk_abs16(dx02), ax13 = pk_abs16(dx13), ay02 = pk_abs16(dy02), ay13 = pk_abs16(dy13);
                    const unsigned inm = (ym >= 0 && ym < h) ? lane_mask : 0u;                 // magnitude is 0 outside the image
                    const unsigned m02 = (ax02 + ay02) & inm, m13 = (ax13 + ay13) & inm;       // <= 2040 per half: plain add
                    M02[k] = m02, M13[k] = m13;
                    ME[k] = hi_lo(dpp_prev_u32(m13), dpp_next_u32(m02));                       // (m of column x-1 | m of column x+4 << 16)
                    // direction class, cv::Canny's fixed-point tests in 16 bits: with t22 = floor(|gx| * 13573 / 2^15)
                    //   |gy| * 2^15 <  |gx| * 13573           <=>  |gy| <= t22         (|gx| > 0; at |gx| = 0 both sides need |gy| = 0,
                    //                                                                   where the magnitude is 0 and nothing is a maximum)
                    //   |gy| * 2^15 >  |gx| * (13573 + 2^16)  <=>  |gy| >  t22 + 2 |gx|
                    // 13573 = 53 * 256 + 5, so t22 = (53 |gx| + ((5 |gx|) >> 8)) >> 7 without leaving 16 bits (|gx| <= 1020)
                    const unsigned t02q = pk_shru(pk_addu(pk_mulu(ax02, 53), pk_shru(pk_mulu(ax02, 5), 8)), 7);
                    const unsigned t13q = pk_shru(pk_addu(pk_mulu(ax13, 53), pk_shru(pk_mulu(ax13, 5), 8)), 7);
                    HZ02[k] = opaque(pk_sar15(pk_sub16(pk_sub16(ay02, t02q), 0x00010001u)));           // |gy| - t22 - 1 < 0
                    HZ13[k] = opaque(pk_sar15(pk_sub16(pk_sub16(ay13, t13q), 0x00010001u)));
                    VT02[k] = opaque(pk_sar15(pk_sub16(pk_addu(t02q, pk_shlu(ax02, 1)), ay02)));        // t22 + 2|gx| - |gy| < 0
                    VT13[k] = opaque(pk_sar15(pk_sub16(pk_addu(t13q, pk_shlu(ax13, 1)), ay13)));
                    NG02[k] = opaque(pk_sar15(dx02 ^ dy02)), NG13[k] = opaque(pk_sar15(dx13 ^ dy13));          // gradient signs differ
                }
                // ---- NMS of row yo = r-4: magnitude rows r-5 (slot k2), r-4 (k1), r-3 (k); class masks of row r-4 (k1) ----
                const int yo = r - 4;
                if (yo >= yb && yo < y_end) {
                    const unsigned mT02 = M02[k2], mT13 = M13[k2], mTE = ME[k2], mC02 = M02[k1], mC13 = M13[k1], mCE = ME[k1],
                                   mB02 = M02[k], mB13 = M13[k], mBE = ME[k];
                    // pixels 0,2: left (m-1, m1), right M13; pixels 1,3: left M02, right (m2, m4); same for the rows above / below
                    const unsigned l02 = lo_lo(mCE, mC13), r13 = hi_hi(mC02, mCE);
                    const unsigned ul02 = lo_lo(mTE, mT13), ur13 = hi_hi(mT02, mTE), dl02 = lo_lo(mBE, mB13), dr13 = hi_hi(mB02, mBE);
                    const unsigned hz02 = HZ02[k1], hz13 = HZ13[k1], vt02 = VT02[k1], vt13 = VT13[k1], ng02 = NG02[k1], ng13 = NG13[k1];
                    // first neighbour (must be strictly smaller): left | up | up-right (signs differ) | up-left
                    const unsigned n1a = bsel(hz02, l02, bsel(vt02, mT02, bsel(ng02, mT13, ul02)));
                    const unsigned n1b = bsel(hz13, mC02, bsel(vt13, mT13, bsel(ng13, ur13, mT02)));
                    // second neighbour: right | down (these two may be equal) | down-left (signs differ) | down-right
                    const unsigned n2a = bsel(hz02, mC13, bsel(vt02, mB02, bsel(ng02, dl02, mB13)));
                    const unsigned n2b = bsel(hz13, r13, bsel(vt13, mB13, bsel(ng13, mB02, dr13)));
                    // m > n1 and (m >= n2 on the horizontal / vertical classes, m > n2 on the diagonals): m - mask adds 1 where mask = 0xFFFF
                    const unsigned ea = pk_minu(pk_subsat(mC02, n1a), pk_subsat(pk_sub16(mC02, hz02 | vt02), n2a));
                    const unsigned eb = pk_minu(pk_subsat(mC13, n1b), pk_subsat(pk_sub16(mC13, hz13 | vt13), n2b));
                    const unsigned ca = pk_minu(pk_shru(mC02, 1), 0x00FF00FFu);
                    const unsigned cb = pk_minu(pk_shru(mC13, 1), 0x00FF00FFu);
                    const unsigned oa = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, ca) * __builtin_bit_cast(u16x2_t, opaque(pk_minu(ea, 0x00010001u))));
                    const unsigned ob = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, cb) * __builtin_bit_cast(u16x2_t, opaque(pk_minu(eb, 0x00010001u))));
                    if (out_lane) *reinterpret_cast<unsigned*>(nm + fo + (size_t)yo * w + x) = oa | (ob << 8);
                }
            }
        }
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += lh[q * 256 + tid];
    if (tot) atomicAdd(&hist[(size_t)s * 256 + tid], tot);
}

// ---- packed-strip front end (front_pack): the same one-pass BGR -> nm pipeline as front_stream, rebuilt around what a
// calibration of the vector pipe (tools/wvalu.hip) and in-kernel clocks (tools/ltime.py) showed about front_stream:
//  * it is bound by vector-instruction issue (196 per 4-pixel row, ~3.5 cycles each with six waves per SIMD), so every
//    instruction that could go went: weights x4 so that gray is byte 2 of the accumulator (no shifts, one v_perm packs a
//    pair), x + 2y and x*6 + y as v_pk_mad_u16 (hipcc emits shift + add), the rounding constant inside a v_add3, LDS
//    histogram addresses by v_mad_u32_u16 straight from the packed halves, row addresses as a scalar base + a per-lane
//    32-bit offset (no 64-bit multiply-adds, no register copies behind the loads), and all tests that only concern the
//    image's border rows / columns moved out of the steady state: border rows are fixed up under scalar branches, border
//    columns exist only in the EDGE instantiation that the first strip and the remainder waves run;
//  * 1280 columns are 320 four-pixel chunks = 5 strips of 62 + 10: the sixth strip used a whole wave for 10 chunks.
//    Remainder chunks of the same band of FIVE frames now share one wave (12 lanes each incl. their two halo lanes), so a
//    band of five frames takes 26 waves instead of 30;
//  * equal-priority waves are served oldest first: the six waves of a SIMD finished one after the other (30 ... 114 us for
//    identical work) and the tail ran at one to three waves per SIMD, where an instruction costs 3.6-5.4 cycles instead of
//    2.5.  A wave now lowers its own priority as it advances (3 -> 2 -> 1 -> 0), so whoever is behind outranks whoever is
//    ahead and all waves of a SIMD finish together; the priority is raised by the kernel's first instruction because a
//    newly placed wave (priority 0) would otherwise starve in its prologue.
struct FrontGeo {
    int S, nb, nfull, rem, G, nc;            // frames, bands, full strips per band, remainder chunks, frames per remainder wave, histogram copies in LDS
};

template <int K>
__device__ __forceinline__ unsigned pk_madk(unsigned a, unsigned c) {       // a * K + c per 16-bit half
    unsigned d;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "n"(K), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pk_min1(unsigned a) {                  // min(a, 1) per half
    unsigned d;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ unsigned mad_lo16x4(unsigned p, unsigned base) {  // (p & 0xFFFF) * 4 + base
    unsigned d;
    asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(d) : "v"(p), "v"(base));
    return d;
}
__device__ __forceinline__ unsigned mad_hi16x4(unsigned p, unsigned base) {  // (p >> 16) * 4 + base
    unsigned d;
    asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(d) : "v"(p), "v"(base));
    return d;
}
__device__ __forceinline__ void lds_add(unsigned byte_addr, unsigned v) {
    asm volatile("ds_add_u32 %0, %1" : : "v"(byte_addr), "v"(v) : "memory");
}

template <bool KEEP_BLUR, int FROWS, bool TIMED, int ABL = 0>
__global__ void __launch_bounds__(64, 5) front_pack(const uint8_t* __restrict__ bgr, int h, int w, FrontGeo geo, uint8_t* __restrict__ blur,
                                                 uint8_t* __restrict__ nm, unsigned* __restrict__ hist) {
    // One wave per workgroup: 5328 equal work items on 256 CUs are 20.8 per CU -- in four-wave workgroups that was 5 or 6
    // workgroups per CU and the CUs with six set the kernel's time (100 us against 79 on the others).
    extern __shared__ unsigned lh[];                                     // geo.nc histogram copies of 256 bins
    __builtin_amdgcn_s_setprio(3);
    unsigned long long tm0 = 0, tr0 = 0, sect = 0;
    if (TIMED) tm0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x;
    for (int i = lane; i < geo.nc * 256; i += 64) lh[i] = 0;
    const int C = w >> 2, per_frame = geo.nb * geo.nfull;
    // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with an L2 of its own.  Neighbouring strips share the
    // 128-byte lines at their seam and neighbouring bands their 8 halo rows, so an XCD takes a CONTIGUOUS eighth of the work
    // items (8 whole frames of the 64) in dispatch order -- placement is for speed only, any mapping is correct.
    int bx = (int)blockIdx.x;
    {
        const int n_full = geo.S * per_frame, per_xcd = n_full >> 3;
        if (bx < per_xcd * 8) bx = (bx & 7) * per_xcd + (bx >> 3);
    }
    const bool full = bx < geo.S * per_frame;
    int band, cidx, s_l, s0, hcopy;
    bool halo, lane_on = true, edge;
    if (full) {
        s0 = __builtin_amdgcn_readfirstlane(bx / per_frame);            // (integer division runs on the vector unit: bring the
        const int item = bx - s0 * per_frame;                            // wave-uniform results back to scalar registers)
        band = __builtin_amdgcn_readfirstlane(item / geo.nfull);
        const int strip = item - band * geo.nfull;
        cidx = 62 * strip - 1 + lane, s_l = s0, halo = lane == 0 || lane == 63, hcopy = lane % geo.nc;
        edge = strip == 0 || 62 * strip + 62 >= C - 1;
    } else {
        const int rb = bx - geo.S * per_frame, grp = __builtin_amdgcn_readfirstlane(rb / geo.nb), gw = geo.rem + 2;
        band = rb - grp * geo.nb;
        const int gi = lane / gw, li = lane - gi * gw, cpg = geo.nc / geo.G;
        s0 = grp * geo.G, s_l = s0 + gi;
        lane_on = gi < geo.G && s_l < geo.S;
        cidx = 62 * geo.nfull - 1 + li, halo = li == 0 || li == gw - 1;
        hcopy = lane_on ? gi * cpg + li % cpg : 0;
        edge = true;
    }
    const bool wave_on = true;
    band = __builtin_amdgcn_readfirstlane(band), s0 = __builtin_amdgcn_readfirstlane(s0);
    edge = __builtin_amdgcn_readfirstlane((int)edge) != 0;
    const bool xin = lane_on && cidx >= 0 && cidx < C;
    const bool out_lane = xin && !halo;
    const bool at_left = xin && cidx == 0, at_right = xin && cidx == C - 1;
    const unsigned lane_mask = xin ? 0xFFFFFFFFu : 0u;
    const unsigned xs = xin ? 4u * (unsigned)cidx : 0u, sl = lane_on ? (unsigned)s_l : (unsigned)s0;
    const unsigned voff_in = sl * ((unsigned)h * (unsigned)w * 3u) + xs * 3u;      // host checked: S*h*w*3 < 2^32
    const unsigned voff_out = sl * ((unsigned)h * (unsigned)w) + xs;
    const unsigned hbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lh + (unsigned)hcopy * 1024u;      // LDS byte address of this lane's histogram copy
    const int yb = band * FROWS;
    if (wave_on && yb < h) {
        const int y_end = (yb + FROWS < h ? yb + FROWS : h);
        const int r_first = yb - 4, r_last = y_end + 3;
        const unsigned pitch = (unsigned)w * 3u;
        auto run = [&](auto edge_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
            unsigned H02[3] = {0, 0, 0}, H13[3] = {0, 0, 0}, U02[3] = {0, 0, 0}, U13[3] = {0, 0, 0};
            unsigned P02[3] = {0, 0, 0}, P13[3] = {0, 0, 0}, PE[3] = {0, 0, 0};
            unsigned M02[3] = {0, 0, 0}, M13[3] = {0, 0, 0}, ME[3] = {0, 0, 0};
            unsigned HZ02[3] = {0, 0, 0}, HZ13[3] = {0, 0, 0}, VT02[3] = {0, 0, 0}, VT13[3] = {0, 0, 0}, NG02[3] = {0, 0, 0},
                     NG13[3] = {0, 0, 0};
            unsigned fa[3], fb[3], fc[3];                                // BGR rows in flight: slot r % 3 holds row r, loaded two rows ahead (three: 12 spilled registers, 137 us)
            // kernel-argument base + one 32-bit offset (row term scalar): a single add and a load with a scalar base
            auto fetch_at = [&](unsigned ys, unsigned& a, unsigned& b, unsigned& c) {
                if (ABL & 2) { a = ys * 0x01010101u + (unsigned)lane * 0x00030201u; b = a ^ 0x05050505u; c = a + 0x00010203u; return; }
                const unsigned* p = reinterpret_cast<const unsigned*>(bgr + (voff_in + ys * pitch));
                a = p[0], b = p[1], c = p[2];
            };
            auto fetch = [&](int yy, unsigned& a, unsigned& b, unsigned& c) {
                fetch_at((unsigned)reflect101_once(yy < r_last ? yy : r_last, h), a, b, c);
            };
            fetch(r_first, fa[0], fb[0], fc[0]);
            fetch(r_first + 1, fa[1], fb[1], fc[1]);
            const unsigned flat_count = 4u * (unsigned)__popcll(__ballot(out_lane));
            // One row of the pipeline; k = r mod 3 is a compile-time ring slot.  STEADY rows lie strictly inside the band and
            // the image: every row test is known (histogram and store on, no border fix-ups, no reflection of the fetched
            // row), which is what takes the scalar unit's ~70 instructions per row out of the loop.
            auto row = [&](auto steady_c, auto k_c, const int r) {
                constexpr bool STEADY = decltype(steady_c)::value;
                constexpr int k = decltype(k_c)::value, k1 = (k + 2) % 3, k2 = (k + 1) % 3;        // slots of rows r-1 and r-2 (= r+1 mod 3)
                // ---- gray row r.  Weights x 4: acc = 4 * (1868 B + 9617 G + 4899 R + 8192), gray = byte 2 of acc ----
                const unsigned a = fa[k], b = fb[k], c = fc[k];          // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                if (STEADY) fetch_at((unsigned)(r + 2), fa[k1], fb[k1], fc[k1]);        // slot (r + 2) % 3 = the one consumed last trip
                else fetch(r + 2, fa[k1], fb[k1], fc[k1]);
                const u16x2_t wbg = {7472, 38468};
                auto acc = [&](unsigned bg_pair, unsigned rr) {
                    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, bg_pair), wbg, (unsigned)__umul24(19596u, rr) + 32768u, false);
                };
                const unsigned a0 = acc(__builtin_amdgcn_perm(0u, a, 0x0C010C00u), (a >> 16) & 255u);
                const unsigned a1 = acc(__builtin_amdgcn_perm(b, a, 0x0C040C03u), (b >> 8) & 255u);
                const unsigned a2 = acc(__builtin_amdgcn_perm(0u, b, 0x0C030C02u), c & 255u);
                const unsigned a3 = acc(__builtin_amdgcn_perm(0u, c, 0x0C020C01u), c >> 24);
                const unsigned G02 = __builtin_amdgcn_perm(a2, a0, 0x0C060C02u), G13 = __builtin_amdgcn_perm(a3, a1, 0x0C060C02u);
                // ---- horizontal 1 4 6 4 1 ----
                unsigned gprev = dpp_prev_u32(hi_hi(G02, G13));          // the previous lane's (g2, g3) = columns x-2, x-1
                unsigned gnext = dpp_next_u32(lo_lo(G02, G13));          // the next lane's (g0, g1) = columns x+4, x+5
                if (EDGE) {
                    const unsigned R21 = hi_lo(G02, G13);                // reflect-101: columns -2,-1 and w,w+1 mirror to (g2, g1)
                    gprev = at_left ? R21 : gprev, gnext = at_right ? R21 : gnext;
                }
                const unsigned L2 = lo_lo(gprev, G02), L1 = hi_lo(gprev, G13);          // (g-2, g0), (g-1, g1)
                const unsigned R2 = hi_lo(G02, gnext), R3 = hi_hi(G13, gnext);          // (g2, g4), (g3, g5)
                H02[k] = pk_madk<6>(G02, pk_madk<4>(L1 + G13, L2 + R2));                // sums <= 4080 per half: plain adds carry nothing across
                H13[k] = pk_madk<6>(G13, pk_madk<4>(G02 + R2, L1 + R3));
                // ---- vertical (1 2 1) twice; blurred row bl = r-2 ----
                U02[k] = pk_madk<2>(H02[k1], H02[k2] + H02[k]);
                U13[k] = pk_madk<2>(H13[k1], H13[k2] + H13[k]);
                const unsigned p02 = pk_shru(pk_madk<2>(U02[k1], U02[k2] + U02[k] + 0x00800080u), 8);
                const unsigned p13 = pk_shru(pk_madk<2>(U13[k1], U13[k2] + U13[k] + 0x00800080u), 8);
                P02[k] = p02, P13[k] = p13;
                const int bl = r - 2;
                if (STEADY || (bl >= yb && bl < y_end)) {                // this wave owns the blurred row: histogram (+ debug copy)
                    // A row whose 248 pixels all have one value (sky, flat surfaces) would be 4 x 62 adds to ONE bin -- the
                    // slowest case for LDS atomics, which serialise per address: one lane adds 248 instead.
                    const unsigned v1 = (unsigned)__builtin_amdgcn_readlane((int)p02, 1);
                    const bool flat = full && (v1 & 0xFFFFu) == (v1 >> 16) && __ballot(out_lane && (p02 != v1 || p13 != v1)) == 0ull;
                    if (KEEP_BLUR && out_lane) *reinterpret_cast<unsigned*>(blur + (voff_out + (unsigned)bl * (unsigned)w)) = p02 | (p13 << 8);
                    if (flat && !(ABL & 1)) {
                        if (lane == 1) lds_add(mad_lo16x4(p02, hbase), flat_count);
                    } else if (out_lane && !(ABL & 1)) {
                        unsigned t0, t1, t2, t3;                         // LDS addresses straight from the packed halves
                        asm volatile("v_mad_u32_u16 %0, %4, 4, %6\n\tv_mad_u32_u16 %1, %4, 4, %6 op_sel:[1,0,0,0]\n\t"
                                     "v_mad_u32_u16 %2, %5, 4, %6\n\tv_mad_u32_u16 %3, %5, 4, %6 op_sel:[1,0,0,0]\n\t"
                                     "ds_add_u32 %0, %7\n\tds_add_u32 %1, %7\n\tds_add_u32 %2, %7\n\tds_add_u32 %3, %7"
                                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p02), "v"(p13), "v"(hbase), "v"(1u) : "memory");
                    }
                }
                {   // blurred neighbours across the lane border: PE = (column x-1 | column x+4 << 16)
                    unsigned dp = dpp_prev_u32(p13), dn = dpp_next_u32(p02);            // .hi = o3 of the previous lane, .lo = o0 of the next
                    if (EDGE) dp = at_left ? (p02 << 16) : dp, dn = at_right ? (p13 >> 16) : dn;     // replicate at the image edge
                    PE[k] = hi_lo(dp, dn);
                }
                // Sobel reads blurred(-1) as blurred(0) and blurred(h) as blurred(h-1) (replicate): rewrite the ring slot
                // when those rows pass -- two scalar tests per border row instead of six selects on every row
                if (!STEADY && bl == 0) {
                    asm volatile("; blurred(-1) := blurred(0)");
                    P02[k1] = P02[k], P13[k1] = P13[k], PE[k1] = PE[k];
                }
                if (!STEADY && bl == h) {
                    asm volatile("; blurred(h) := blurred(h-1)");
                    P02[k] = P02[k1], P13[k] = P13[k1], PE[k] = PE[k1];
                }
                // ---- Sobel of row ym = r-3 from blurred rows r-4 (slot k2), r-3 (k1), r-2 (k) ----
                const int ym = r - 3;
                unsigned long long ts0 = 0;
                if (TIMED) ts0 = __builtin_amdgcn_s_memtime();
                {
                    const unsigned t02 = P02[k2], t13 = P13[k2], te = PE[k2], b02 = P02[k], b13 = P13[k], be = PE[k];
                    const unsigned c02 = P02[k1], c13 = P13[k1], ce = PE[k1];
                    const unsigned v02 = pk_madk<2>(c02, t02 + b02), v13 = pk_madk<2>(c13, t13 + b13), ve = pk_madk<2>(ce, te + be);
                    const unsigned d02 = pk_sub16(b02, t02), d13 = pk_sub16(b13, t13), de = pk_sub16(be, te);
                    const unsigned dx02 = pk_sub16(v13, lo_lo(ve, v13));                           // V[k+1] - V[k-1] for pixels 0, 2
                    const unsigned dx13 = pk_sub16(hi_hi(v02, ve), v02);                           //                  for pixels 1, 3
                    const unsigned dy02 = pk_add16(pk_madk<2>(d02, lo_lo(de, d13)), d13);
                    const unsigned dy13 = pk_add16(pk_madk<2>(d13, d02), hi_hi(d02, de));
                    const unsigned ax02 = pk_abs16(dx02), ax13 = pk_abs16(dx13), ay02 = pk_abs16(dy02), ay13 = pk_abs16(dy13);
                    unsigned m02 = ax02 + ay02, m13 = ax13 + ay13;                                 // <= 2040 per half: plain add
                    if (EDGE) m02 &= lane_mask, m13 &= lane_mask;                                  // magnitude is 0 outside the image's columns
                    if (!STEADY && (ym < 0 || ym >= h)) {                                          // ... and outside its rows
                        asm volatile("; magnitude row outside the image");
                        m02 = 0, m13 = 0;
                    }
                    M02[k] = m02, M13[k] = m13;
                    ME[k] = hi_lo(dpp_prev_u32(m13), dpp_next_u32(m02));                           // (m of column x-1 | m of column x+4 << 16)
                    // direction classes: see front_stream
                    const unsigned t02q = pk_shru(pk_madk<53>(ax02, pk_shru(pk_mulu(ax02, 5), 8)), 7);
                    const unsigned t13q = pk_shru(pk_madk<53>(ax13, pk_shru(pk_mulu(ax13, 5), 8)), 7);
                    HZ02[k] = opaque(pk_sar15(pk_add16(ay02, ~t02q)));                              // |gy| - t22 - 1 < 0
                    HZ13[k] = opaque(pk_sar15(pk_add16(ay13, ~t13q)));
                    VT02[k] = opaque(pk_sar15(pk_sub16(pk_madk<2>(ax02, t02q), ay02)));             // t22 + 2|gx| - |gy| < 0
                    VT13[k] = opaque(pk_sar15(pk_sub16(pk_madk<2>(ax13, t13q), ay13)));
                    NG02[k] = opaque(pk_sar15(dx02 ^ dy02)), NG13[k] = opaque(pk_sar15(dx13 ^ dy13));              // gradient signs differ
                }
                if (TIMED) {
                    asm volatile("" :: "v"(NG13[k]), "v"(VT13[k]), "v"(HZ13[k]), "v"(ME[k]));
                    sect += __builtin_amdgcn_s_memtime() - ts0;
                }
                // ---- NMS of row yo = r-4: magnitude rows r-5 (slot k2), r-4 (k1), r-3 (k); class masks of row r-4 (k1) ----
                const int yo = r - 4;
                if (STEADY || (yo >= yb && yo < y_end)) {
                    const unsigned mT02 = M02[k2], mT13 = M13[k2], mTE = ME[k2], mC02 = M02[k1], mC13 = M13[k1], mCE = ME[k1],
                                   mB02 = M02[k], mB13 = M13[k], mBE = ME[k];
                    const unsigned l02 = lo_lo(mCE, mC13), r13 = hi_hi(mC02, mCE);
                    const unsigned ul02 = lo_lo(mTE, mT13), ur13 = hi_hi(mT02, mTE), dl02 = lo_lo(mBE, mB13), dr13 = hi_hi(mB02, mBE);
                    const unsigned hz02 = HZ02[k1], hz13 = HZ13[k1], vt02 = VT02[k1], vt13 = VT13[k1], ng02 = NG02[k1], ng13 = NG13[k1];
                    const unsigned n1a = bsel(hz02, l02, bsel(vt02, mT02, bsel(ng02, mT13, ul02)));
                    const unsigned n1b = bsel(hz13, mC02, bsel(vt13, mT13, bsel(ng13, ur13, mT02)));
                    const unsigned n2a = bsel(hz02, mC13, bsel(vt02, mB02, bsel(ng02, dl02, mB13)));
                    const unsigned n2b = bsel(hz13, r13, bsel(vt13, mB13, bsel(ng13, mB02, dr13)));
                    const unsigned ea = pk_minu(pk_subsat(mC02, n1a), pk_subsat(pk_sub16(mC02, hz02 | vt02), n2a));
                    const unsigned eb = pk_minu(pk_subsat(mC13, n1b), pk_subsat(pk_sub16(mC13, hz13 | vt13), n2b));
                    const unsigned ca = pk_minu(pk_shru(mC02, 1), 0x00FF00FFu);
                    const unsigned cb = pk_minu(pk_shru(mC13, 1), 0x00FF00FFu);
                    const unsigned oa = pk_mulv(ca, pk_min1(ea)), ob = pk_mulv(cb, pk_min1(eb));
                    if ((ABL & 4) ? (out_lane && (oa ^ ob) == 0x12345678u) : out_lane) *reinterpret_cast<unsigned*>(nm + (voff_out + (unsigned)yo * (unsigned)w)) = oa | (ob << 8);
                }
            };
            const int n_trips = (r_last - r_first + 3) / 3, trip_p2 = r_first + 3 * (n_trips / 2),
                      trip_p1 = r_first + 3 * ((n_trips * 4) / 5), trip_p0 = r_first + 3 * (n_trips - 1);
            int r0 = r_first;
            auto trip = [&](auto steady_c, const int rr) {
                if (rr == trip_p2) __builtin_amdgcn_s_setprio(2);
                if (rr == trip_p1) __builtin_amdgcn_s_setprio(1);
                if (rr == trip_p0) __builtin_amdgcn_s_setprio(0);
                row(steady_c, std::integral_constant<int, 0>{}, rr);
                row(steady_c, std::integral_constant<int, 1>{}, rr + 1);
                row(steady_c, std::integral_constant<int, 2>{}, rr + 2);
            };
            // rows r0 .. r0+2 are STEADY when yb + 4 <= r0 (store and histogram on) and r0 + 2 <= min(y_end + 1, h - 3) (still on,
            // fetched row r + 2 inside the image); the first three trips fill the pipeline
            const int r_hi = (y_end + 1 < h - 3 ? y_end + 1 : h - 3), r_lo = r_first + 9;
#pragma nounroll
            for (;;) {
                const int rr = __builtin_amdgcn_readfirstlane(r0);       // keeps the row counter, and every test on it, on the scalar unit
                if (rr > r_last) break;
                if (rr >= r_lo && rr + 2 <= r_hi) trip(std::true_type{}, rr);
                else trip(std::false_type{}, rr);
                r0 = rr + 3;
            }
        };
        if (edge) run(std::true_type{});
        else run(std::false_type{});
    }
    if (TIMED) {
        const unsigned long long tm1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(blur) + (size_t)bx * 4;
            dbg[0] = tm1 - tm0, dbg[1] = tr1 - tr0, dbg[2] = tr0;
            dbg[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
                     ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 15) << 32) | (sect << 36);
        }
    }
    // the wave's own LDS operations complete in order: its histogram is final here
    if (full) {
        for (int bin = lane; bin < 256; bin += 64) {
            unsigned tot = 0;
            for (int q = 0; q < geo.nc; ++q) tot += lh[q * 256 + bin];
            if (tot) atomicAdd(&hist[(size_t)s0 * 256 + bin], tot);
        }
    } else {
        const int cpg = geo.nc / geo.G;
        for (int gi = 0; gi < geo.G && s0 + gi < geo.S; ++gi)
            for (int bin = lane; bin < 256; bin += 64) {
                unsigned tot = 0;
                for (int q = 0; q < cpg; ++q) tot += lh[(gi * cpg + q) * 256 + bin];
                if (tot) atomicAdd(&hist[(size_t)(s0 + gi) * 256 + bin], tot);
            }
    }
}

// candidate / strong bits of 16 map bytes.  NM: the fused front end's non-maximum-suppressed magnitudes against the
// halved thresholds; otherwise the {0 weak, 1 none, 2 strong} codes of the two-pass kernels.
// NM, four bytes at a time: byte > t for a threshold t <= 127 (the halved Canny thresholds) is bit 7 of ((b & 0x7F) + 0x7F - t) | b -- no
// carry leaves a byte -- and a multiply gathers the four bits: 7 operations per word and threshold instead of ~5 per byte.
__device__ __forceinline__ unsigned gt_nibble(unsigned x, unsigned x7, unsigned k) {      // x7 = x & 0x7F7F7F7F, k = (0x7F - t) * 0x01010101
    const unsigned m = ((x7 + k) | x) & 0x80808080u;
    return (((m >> 7) * 0x00204081u) >> 21) & 0xFu;                                          // bits 0, 8, 16, 24 -> bits 21 .. 24 of the product
}
template <bool NM>
__device__ __forceinline__ void map_bits(const uint4& v, int lo2, int hi2, unsigned& cand, unsigned& strong) {
    if (NM) {
        const unsigned kl = (unsigned)(0x7F - lo2) * 0x01010101u, kh = (unsigned)(0x7F - hi2) * 0x01010101u;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        cand = 0, strong = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned x7 = w[q] & 0x7F7F7F7Fu;
            cand |= gt_nibble(w[q], x7, kl) << (4 * q), strong |= gt_nibble(w[q], x7, kh) << (4 * q);
        }
        return;
    }
    const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
    cand = 0, strong = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) cand |= (b[k] != 1 ? 1u : 0u) << k, strong |= (b[k] == 2 ? 1u : 0u) << k;
}
template <bool NM>
__device__ __forceinline__ bool map_cand(uint8_t v, int lo2) { return NM ? (int)v > lo2 : v != 1; }
__device__ __forceinline__ void half_thresholds(const double* thr, int s, int& lo2, int& hi2) {
    int lo = (int)thr[(size_t)s * 4], hi = (int)thr[(size_t)s * 4 + 1];
    if (lo > hi) { const int q = lo; lo = hi; hi = q; }
    lo2 = lo >> 1, hi2 = hi >> 1;
}

// Chunk index inside a frame -> (row, first column) without an integer division: float reciprocal, then the
// remainder puts an off-by-one quotient right (exact for ci < 2^24 chunks = 268 M pixels per frame).
__device__ __forceinline__ void chunk_xy(unsigned ci, int cw, float rcw, int& y, int& xb) {
    unsigned q = (unsigned)((float)ci * rcw);
    int r = (int)(ci - q * (unsigned)cw);
    if (r < 0) --q, r += cw;
    else if (r >= cw) ++q, r -= cw;
    y = (int)q, xb = r * 16;
}

// ---- hysteresis, tiled: components are first resolved inside 16 x 256 tiles with the labels in LDS (a union there
// is a few LDS round trips; in global memory it is a chain of dependent atomics that large components serialise on),
// every candidate is then pointed at its tile-local root in the global label array, and only the links that cross a
// tile border are made with global unions.  Same label order (strong < weak), so the verdict "root is strong" and
// with it the edge map are unchanged.  Tile height: 64 rows 90 us, 32 rows 52 us, 16 rows 36 us per 64 frames (LDS per
// workgroup sets the occupancy, and most tiles hold no candidate at all); the border pass grows from 13 to 17 us.
constexpr int CT_R = 16, CT_C = 256, CT_CH = CT_C / 16, CT_NCH = CT_R * CT_CH, CT_PX = CT_R * CT_C;
constexpr int TE_WORDS = 20;        // per tile: [0..7] top row bits, [8..15] bottom row bits, [16] first column (bit r = row r), [17] last column

__device__ __forceinline__ unsigned lds_find(unsigned* lab, unsigned v) {
    unsigned i = v & 0x7fffffffu;
    unsigned cur = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((cur & 0x7fffffffu) != i) {
        i = cur & 0x7fffffffu;
        cur = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return cur;
}
__device__ __forceinline__ void lds_union(unsigned* lab, unsigned a, unsigned b) {
    for (;;) {
        unsigned ra = lds_find(lab, a), rb = lds_find(lab, b);
        if (ra == rb) return;
        if (ra < rb) { const unsigned q = ra; ra = rb; rb = q; }
        const unsigned old = atomicMin(&lab[ra & 0x7fffffffu], rb);
        if (old == ra) return;
        a = old, b = rb;
    }
}
// Which links a chunk's pixels have to make, from candidate bits alone.  C / U: columns -1 .. 16 of the pixel's row
// and of the row above as bits 0 .. 17.  Pixels of one horizontal run are one component already (and so are the
// runs of the row above, by whoever owns that row), so a link that the left or right neighbour makes as well is
// skipped: one link per pair of touching runs.  out[0] left (pixel 0 only), [1] up-left, [2] up, [3] up-right.
__device__ __forceinline__ void ccl_links(unsigned C, unsigned U, unsigned out[4]) {
    const unsigned cand = (C >> 1) & 0xFFFFu;
    const unsigned Lm = C & 0xFFFFu, Rm = (C >> 2) & 0xFFFFu, ULm = U & 0xFFFFu, UCm = (U >> 1) & 0xFFFFu, URm = (U >> 2) & 0xFFFFu;
    out[0] = cand & Lm & 1u;
    out[1] = cand & ~UCm & ULm & ~Lm;
    out[2] = cand & UCm & ~(Lm & ULm);
    out[3] = cand & ~UCm & URm & ~Rm;
}

// A workgroup takes CT_STACK vertically adjacent tiles, all map chunks loaded up front, the non-empty ones resolved one
// after the other in the same LDS arrays.  Measured at 720p, 64 frames: one tile per workgroup 37 us, four 53 us --
// the non-empty tiles (lane markings, vehicle outlines) sit above each other and then run serially -- so CT_STACK = 1.
// (Four tiles a quarter of the frame apart, so that a workgroup's tiles are not non-empty together: 47 us -- the kernel's time
// is the non-empty tiles' serial LDS work, not its loads.)
constexpr int CT_STACK = 1;
struct BitBox {                    // the ROI's chunk box in the bit maps: first chunk column / row, chunks per row, rows, 32-bit words per row
    int bx0, by0, bcw, bch, pw;
};
template <bool NM>
__global__ void __launch_bounds__(256) ccl_tile_kernel(const uint8_t* __restrict__ map_all, int h, int w,
                                                       const double* __restrict__ thr, unsigned* __restrict__ labels_all,
                                                       unsigned* __restrict__ tedge_all, const int* __restrict__ roi_tab,
                                                       uint16_t* __restrict__ rbits_all, BitBox bb) {
    __shared__ unsigned lab[CT_PX];               // local label: pixel index inside the tile (row * 256 + column), bit 31 = weak
    __shared__ unsigned cm[CT_NCH];               // per 16-pixel chunk: candidate bits | strong bits << 16
    static_assert(CT_NCH == 256, "one chunk per thread and tile");
    const int tid = threadIdx.x, x0 = blockIdx.x * CT_C, s = blockIdx.z;
    const uint8_t* m = map_all + (size_t)s * h * w;
    unsigned* glab = labels_all + (size_t)s * h * w;
    int lo2 = 0, hi2 = 0;
    if (NM) half_thresholds(thr, s, lo2, hi2);
    const int c = tid, r = c / CT_CH, cc = c % CT_CH;
    const int x = x0 + cc * 16;
    uint4 vq[CT_STACK];
#pragma unroll
    for (int t = 0; t < CT_STACK; ++t) {
        const int y = ((int)blockIdx.y * CT_STACK + t) * CT_R + r;
        const unsigned none = NM ? 0u : 0x01010101u;
        vq[t] = make_uint4(none, none, none, none);
        if (y < h && x < w) vq[t] = *reinterpret_cast<const uint4*>(m + (size_t)y * w + x);      // w % 16 == 0
    }
#pragma unroll
    for (int t = 0; t < CT_STACK; ++t) {
        const int y0 = ((int)blockIdx.y * CT_STACK + t) * CT_R;
        if (y0 >= h) break;
        unsigned cand, strong;
        map_bits<NM>(vq[t], lo2, hi2, cand, strong);
        if (roi_tab) {                 // the chunk's candidates inside the ROI, one bit per pixel, for the resolve pass (every chunk of
            const int yy = y0 + r, cx = x >> 4;      // the box writes its half-word, also those of tiles without any candidate)
            if (yy >= bb.by0 && yy < bb.by0 + bb.bch && cx >= bb.bx0 && cx < bb.bx0 + bb.bcw) {
                const int lo = roi_tab[2 * yy] - x, hi = roi_tab[2 * yy + 1] - x;        // ROI columns [xl, xr] relative to the chunk
                unsigned cm16 = 0;
                if (hi >= 0 && lo <= 15 && lo <= hi) cm16 = (0xFFFFu >> (15 - (hi > 15 ? 15 : hi))) & (0xFFFFu << (lo < 0 ? 0 : lo));
                rbits_all[((size_t)s * bb.bch + (yy - bb.by0)) * (2 * bb.pw) + (cx - bb.bx0)] = (uint16_t)(cand & cm16);
            }
        }
        unsigned* te = tedge_all + (((size_t)s * gridDim.y * CT_STACK + (y0 / CT_R)) * gridDim.x + blockIdx.x) * TE_WORDS;
        if (__syncthreads_or(cand != 0 ? 1 : 0) == 0) {               // no candidate in the tile (also fences the previous tile's reads)
            if (tid < 18) te[tid] = 0;
            continue;
        }
        cm[c] = cand | strong << 16;
        {   // own label, or the smallest label of the horizontal run inside the chunk (first strong pixel, else first pixel)
            unsigned rest = cand;
            while (rest) {
                const int a = __ffs((int)rest) - 1;
                const unsigned run = rest & ~(rest + (1u << a));
                rest &= ~run;
                const unsigned sr = strong & run;
                const int rp = sr ? __ffs((int)sr) - 1 : a;
                const unsigned rep = (unsigned)(c * 16 + rp) | (sr ? 0u : 0x80000000u);
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if ((run >> k) & 1u) lab[c * 16 + k] = rep;
            }
        }
        __syncthreads();
        // the tile's edge candidate bits for the border pass (which then reads 72 bytes per tile instead of whole cache lines of the
        // map for single bytes at the tile seams: 0.9 B/px of fetches)
        if (tid < 16) {
            const int base = tid < 8 ? 2 * tid : (CT_R - 1) * CT_CH + 2 * (tid - 8);
            te[tid] = (cm[base] & 0xFFFFu) | (cm[base + 1] << 16);
        } else if (tid < 18) {
            unsigned bits = 0;
#pragma unroll
            for (int rr = 0; rr < CT_R; ++rr)
                bits |= (tid == 16 ? (cm[rr * CT_CH] & 1u) : ((cm[rr * CT_CH + CT_CH - 1] >> 15) & 1u)) << rr;
            te[tid] = bits;
        }
        if (cand) {
            // neighbours outside the tile count as absent here: the border pass makes those links
            const unsigned Lb = cc > 0 ? (cm[c - 1] >> 15) & 1u : 0u, Rb = cc < CT_CH - 1 ? cm[c + 1] & 1u : 0u;
            unsigned ucand = 0, ULb = 0, URb = 0;
            if (r > 0) {
                ucand = cm[c - CT_CH] & 0xFFFFu;
                ULb = cc > 0 ? (cm[c - CT_CH - 1] >> 15) & 1u : 0u;
                URb = cc < CT_CH - 1 ? cm[c - CT_CH + 1] & 1u : 0u;
            }
            unsigned link[4];
            ccl_links((cand << 1) | Lb | (Rb << 17), (ucand << 1) | ULb | (URb << 17), link);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned bits = link[q];
                while (bits) {
                    const int k = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    const unsigned me = (unsigned)(c * 16 + k);
                    lds_union(lab, me, q == 0 ? me - 1u : me - (unsigned)CT_C + (unsigned)q - 2u);
                }
            }
        }
        __syncthreads();
        // every candidate -> the global label of its tile-local root
        unsigned bits = cand;
        unsigned* out = glab + (size_t)(y0 + r) * w + x;
        while (bits) {
            const int k = __ffs((int)bits) - 1;
            bits &= bits - 1;
            const unsigned root = lds_find(lab, (unsigned)(c * 16 + k));
            const unsigned rl = root & 0x7fffffffu;
            out[k] = (unsigned)((y0 + (int)(rl >> 8)) * w + x0 + (int)(rl & 255u)) | (root & 0x80000000u);
        }
    }
}

// Links across tile borders, with global unions.  blockIdx.y < nbh: the horizontal border above row (blockIdx.y+1)*16,
// one thread per column; else the vertical border left of column (blockIdx.y-nbh+1)*256, one thread per row.  Same
// skipping rules as inside the tiles, with every neighbour's true candidate bit -- read from the edge masks the tile pass
// left per tile, not from the map (round 2: 52.8 MB of fetches per 64 frames, mostly whole lines for single bytes at the seams;
// now 6.1 MB).  One thread per border PIXEL on purpose: a thread per 32-bit mask word, which walks its set bits, measured 24 us
// against 15 -- the pass's time is the unions' dependent global atomics, which want to be spread over threads.
__global__ void __launch_bounds__(256) ccl_border_kernel(int h, int w, int nbh, const unsigned* __restrict__ tedge_all, int tiles_x,
                                                         int tiles_y, unsigned* __restrict__ labels_all) {
    const int s = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    unsigned* lab = labels_all + (size_t)s * h * w;
    const unsigned* te = tedge_all + (size_t)s * tiles_y * tiles_x * TE_WORDS;
    // candidate bit of (row `which` of tile row ty: 0 = its top row, 1 = its bottom row; column x), 0 outside the image
    auto row_bit = [&](int ty, int which, int x) -> bool {
        if (x < 0 || x >= w) return false;
        const unsigned wd = te[((size_t)ty * tiles_x + (x >> 8)) * TE_WORDS + which * 8 + ((x & 255) >> 5)];
        return (wd >> (x & 31)) & 1u;
    };
    if ((int)blockIdx.y < nbh) {
        const int ty = (int)blockIdx.y + 1, y = ty * CT_R, x = i;     // the border above tile row ty
        if (x >= w || !row_bit(ty, 0, x)) return;
        const bool L = row_bit(ty, 0, x - 1), R = row_bit(ty, 0, x + 1);
        const bool UL = row_bit(ty - 1, 1, x - 1), UC = row_bit(ty - 1, 1, x), UR = row_bit(ty - 1, 1, x + 1);
        const unsigned me = (unsigned)(y * w + x), up = me - (unsigned)w;
        if (UC) {
            if (!(L && UL)) uf_union(lab, me, up);
        } else {
            if (UL && !L) uf_union(lab, me, up - 1u);
            if (UR && !R) uf_union(lab, me, up + 1u);
        }
    } else {
        const int tx = (int)blockIdx.y - nbh + 1, x = tx * CT_C, y = i;       // A = (y, x), B = (y, x - 1) on the other side
        if (y >= h) return;
        const int ty = y / CT_R, r = y - ty * CT_R;
        const unsigned lcol = te[((size_t)ty * tiles_x + tx) * TE_WORDS + 16], rcol = te[((size_t)ty * tiles_x + tx - 1) * TE_WORDS + 17];
        const bool A = (lcol >> r) & 1u, B = (rcol >> r) & 1u;
        const unsigned a = (unsigned)(y * w + x);
        if (A && B) uf_union(lab, a, a - 1u);
        if (r == 0) return;                                               // the horizontal pass owns the links to the row above
        const bool Au = (lcol >> (r - 1)) & 1u, Bu = (rcol >> (r - 1)) & 1u;       // (y - 1, x), (y - 1, x - 1)
        if (A && !B && Bu && !Au) uf_union(lab, a, a - (unsigned)w - 1u);       // A's up-left
        if (B && !A && Au && !Bu) uf_union(lab, a - 1u, a - (unsigned)w);       // B's up-right
    }
}

// Resolve every candidate's component (strong root = an edge), apply the ROI, count the ROI edges per row.
// A thread takes FCK 16-pixel chunks a workgroup-stride apart (coalesced per trip, loads in flight together).
// Only the box of chunks [bx0, bx0 + bcw) x [by0, by0 + bch) is visited: the whole frame when the pre-ROI edge map is
// wanted or the ROI is caller-defined, else the bounding box of the default trapezoid -- outside it the masked map
// is never written by anybody (it is zero since av_lane_workspace_init and the Hough stage only erases).
constexpr int FCK = 1;        // one chunk per thread: a thread's component look-ups are a dependent chain, more of them per thread only lengthens it
template <bool NM>
__global__ void __launch_bounds__(256) finalize_fast(const uint8_t* __restrict__ map_all, int h, int w,
                                                     const double* __restrict__ thr, unsigned* __restrict__ labels_all, Roi roi,
                                                     const int* __restrict__ roi_rows, uint8_t* __restrict__ edges_all,
                                                     uint8_t* __restrict__ masked_all, int* __restrict__ rowcnt, int bx0,
                                                     int by0, int bcw, int bch) {
    const int s = blockIdx.y;
    const float rcw = 1.0f / (float)bcw;
    const unsigned total = (unsigned)(bch * bcw);
    const unsigned c0 = blockIdx.x * (256u * FCK) + threadIdx.x;
    const size_t fo = (size_t)s * h * w;
    const uint8_t* m = map_all + fo;
    unsigned* lab = labels_all + fo;
    int lo2 = 0, hi2 = 0;
    if (NM) half_thresholds(thr, s, lo2, hi2);
    uint4 cur4[FCK];
    int yy[FCK], xx0[FCK];
#pragma unroll
    for (int g = 0; g < FCK; ++g) {
        const unsigned ci = c0 + (unsigned)g * 256u;
        const unsigned none = NM ? 0u : 0x01010101u;
        cur4[g] = make_uint4(none, none, none, none);
        yy[g] = 0, xx0[g] = 0;
        if (ci < total) {
            int y, xb;
            chunk_xy(ci, bcw, rcw, y, xb);
            yy[g] = by0 + y, xx0[g] = bx0 * 16 + xb;
            cur4[g] = *reinterpret_cast<const uint4*>(m + (size_t)yy[g] * w + xx0[g]);
        }
    }
#pragma unroll
    for (int g = 0; g < FCK; ++g) {
        const unsigned ci = c0 + (unsigned)g * 256u;
        if (ci >= total) break;
        uint4 e4 = make_uint4(0, 0, 0, 0), k4 = e4;
        unsigned candm, strongm;
        map_bits<NM>(cur4[g], lo2, hi2, candm, strongm);
        const int y = yy[g], xb = xx0[g];
        if (candm) {
            int xl, xr, cnt = 0;
            roi_bounds(roi, h, y, roi_rows, xl, xr);
            uint8_t* e = reinterpret_cast<uint8_t*>(&e4);
            uint8_t* k = reinterpret_cast<uint8_t*>(&k4);
            bool run_on = false, keep = false;               // adjacent candidates share their component: one find per run
            for (int q = 0; q < 16; ++q) {
                if (!((candm >> q) & 1u)) {
                    run_on = false;
                    continue;
                }
                const int xx = xb + q;
                if (!run_on) keep = !(uf_find(lab, (unsigned)(y * w + xx)) >> 31), run_on = true;
                if (keep) {
                    e[q] = 255;
                    if (xx >= xl && xx <= xr) k[q] = 255, ++cnt;
                }
            }
            if (cnt) atomicAdd(&rowcnt[(size_t)s * h + y], cnt);
        }
        const size_t off = fo + (size_t)y * w + xb;
        if (edges_all) *reinterpret_cast<uint4*>(edges_all + off) = e4;
        *reinterpret_cast<uint4*>(masked_all + off) = k4;
    }
}

// The same verdict from the tile pass's ROI candidate bits (production shape: default ROI, no debug edge map): a thread per 32-bit
// word of the box.  Neither the map nor the masked byte map is touched -- the resolve pass used to re-read 1 B/px of the box and
// write 1 B/px of masked map that the compaction read back; now both passes move one BIT per box pixel.  Adjacent candidates
// share their component: one find per run of set bits.
__global__ void __launch_bounds__(256) resolve_bits_kernel(int h, int w, unsigned* __restrict__ labels_all,
                                                          const unsigned* __restrict__ rbits_all, unsigned* __restrict__ kbits_all,
                                                          int* __restrict__ rowcnt, BitBox bb) {
    const int s = blockIdx.y;
    const unsigned i = blockIdx.x * 256u + threadIdx.x, total = (unsigned)(bb.bch * bb.pw);
    if (i >= total) return;
    const size_t wi = (size_t)s * total + i;
    unsigned rest = rbits_all[wi], keep = 0;
    if (rest) {
        unsigned* lab = labels_all + (size_t)s * h * w;
        const unsigned row = i / (unsigned)bb.pw, wx = i - row * (unsigned)bb.pw;
        const int y = bb.by0 + (int)row, xw = bb.bx0 * 16 + 32 * (int)wx;
        while (rest) {
            const int a = __ffs((int)rest) - 1;
            const unsigned run = rest & ~(rest + (1u << a));
            rest &= ~run;
            if (!(uf_find(lab, (unsigned)(y * w + xw + a)) >> 31)) keep |= run;
        }
        if (keep) atomicAdd(&rowcnt[(size_t)s * h + y], __popc(keep));
    }
    kbits_all[wi] = keep;
}

// ---- L4a: row-major list of edge points -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) compact_kernel(const uint8_t* __restrict__ masked, int h, int w,
                                                      const int* __restrict__ rowcnt, unsigned* __restrict__ nz,
                                                      int* __restrict__ npts) {
    __shared__ int red[4];
    __shared__ int base_s;
    const int s = blockIdx.y, y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int* rc = rowcnt + (size_t)s * h;
    if (rc[y] == 0 && y != h - 1) return;
    int part = 0;
    for (int i = tid; i < y; i += 256) part += rc[i];
    part = wave_sum_i(part);
    if (lane == 0) red[wid] = part;
    __syncthreads();
    if (tid == 0) base_s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    int base = base_s;
    if (y == h - 1 && tid == 0) npts[s] = base + rc[y];
    const uint8_t* row = masked + ((size_t)s * h + y) * w;
    unsigned* out = nz + (size_t)s * h * w;
    for (int x0 = 0; x0 < w; x0 += 256) {
        const int x = x0 + tid;
        const bool on = x < w && row[x] != 0;
        const unsigned long long bal = __ballot(on);
        __syncthreads();
        if (lane == 0) red[wid] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int q = 0; q < wid; ++q) off += red[q];
        const int tot = red[0] + red[1] + red[2] + red[3];
        if (on) out[base + off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned)x | ((unsigned)y << 16);
        base += tot;
    }
}

// The same list for the chunk box the resolve pass visited (w % 16 == 0): a few workgroups per frame, each scans the
// box's per-row counts in LDS (redundantly), then their waves share the rows of the box, a lane per 16-pixel chunk: non-zero bytes ->
// bit mask -> wave prefix of the pop-counts -> the points, row-major.  (The per-row kernel above starts one workgroup
// per image row, each summing all the counts before it: 46 080 workgroups and 30 us per 64 frames at 720p.)
constexpr int CB_ROWS = 8192;                   // rows the scan holds (the box is 288 rows for the default ROI at 720p)
__global__ void __launch_bounds__(1024) compact_box_kernel(const uint8_t* __restrict__ masked, int h, int w,
                                                           const int* __restrict__ rowcnt, unsigned* __restrict__ nz,
                                                           int* __restrict__ npts, int bx0, int by0, int bcw, int bch,
                                                           const unsigned* __restrict__ kbits_all, int pw,
                                                           int* __restrict__ prep_accum, size_t accum_stride, int prep_words,
                                                           int* __restrict__ prep_fallback) {
    // prep_accum (or null): what hough_prep_kernel does for the sharded Hough kernel that follows in the same call -- this frame's
    // exchange words and fallback flag cleared here instead of by a launch of its own
    if (prep_accum && blockIdx.y == 0) {
        if ((int)threadIdx.x < prep_words) prep_accum[(size_t)blockIdx.x * accum_stride + threadIdx.x] = 0;
        if (threadIdx.x == 0) prep_fallback[blockIdx.x] = 0;
    }
    // kbits_all: the kept-edge bit map of the box (resolve_bits_kernel), pw words per row -- a lane per 32-pixel word; else the
    // masked byte map, a lane per 16-pixel chunk
    extern __shared__ int cb_base[];                        // [bch] exclusive prefix of the box rows' counts
    __shared__ int wtot[16];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gw = (int)blockIdx.y * 16 + wid, nw = (int)gridDim.y * 16;      // this wave among the frame's waves (every workgroup repeats the scan)
    const int* rc = rowcnt + (size_t)s * h + by0;
    const int per = (bch + 1023) / 1024;                    // consecutive rows per thread (<= 8)
    int loc[8], sum = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = tid * per + q;
        loc[q] = (q < per && i < bch) ? rc[i] : 0;
        sum += loc[q];
    }
    // exclusive scan of the 1024 per-thread sums: inside each wave by shuffles, then over the 16 wave totals
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
    }
    if (lane == 63) wtot[wid] = inc;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wid; ++q) woff += wtot[q];
    int run = woff + inc - sum;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = tid * per + q;
        if (q < per && i < bch) cb_base[i] = run;
        run += loc[q];
    }
    if (tid == 1023 && blockIdx.y == 0) npts[s] = run;       // thread 1023's running total = all points of the frame
    __syncthreads();
    unsigned* out = nz + (size_t)s * h * w;
    for (int i = gw; i < bch; i += nw) {
        if (rc[i] == 0) continue;                           // wave-uniform
        const int y = by0 + i;
        int base = cb_base[i];
        if (kbits_all) {
            const unsigned* krow = kbits_all + ((size_t)s * bch + i) * pw;
            for (int c0 = 0; c0 < pw; c0 += 64) {
                const int c = c0 + lane;
                unsigned bits = c < pw ? krow[c] : 0u;
                const int n = __popc(bits);
                int pre = n;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(pre, d, 64);
                    if (lane >= d) pre += t;
                }
                const int tot = __shfl(pre, 63, 64);
                int o = base + pre - n;
                const unsigned xb = (unsigned)(bx0 * 16 + c * 32);
                while (bits) {
                    const int k = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    out[o++] = (xb + (unsigned)k) | ((unsigned)y << 16);
                }
                base += tot;
            }
            continue;
        }
        const uint8_t* row = masked + ((size_t)s * h + y) * w + (size_t)bx0 * 16;
        uint4 vnext = make_uint4(0, 0, 0, 0);
        if (lane < bcw) vnext = *reinterpret_cast<const uint4*>(row + (size_t)lane * 16);
        for (int c0 = 0; c0 < bcw; c0 += 64) {
            const int c = c0 + lane;
            const uint4 v = vnext;
            vnext = make_uint4(0, 0, 0, 0);
            if (c + 64 < bcw) vnext = *reinterpret_cast<const uint4*>(row + (size_t)(c + 64) * 16);      // next trip in flight
            const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) bits |= (b[k] != 0 ? 1u : 0u) << k;
            const int n = __popc(bits);
            int pre = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(pre, d, 64);
                if (lane >= d) pre += t;
            }
            const int tot = __shfl(pre, 63, 64);
            int o = base + pre - n;
            const unsigned xb = (unsigned)((bx0 + c) * 16);
            while (bits) {
                const int k = __ffs((int)bits) - 1;
                bits &= bits - 1;
                out[o++] = (xb + (unsigned)k) | ((unsigned)y << 16);
            }
            base += tot;
        }
    }
}

// ---- L4b: progressive probabilistic Hough (cv::HoughLinesProbabilistic) --------------------------------------
struct HoughCfg {
    int threshold, line_length, line_gap, max_segments;
};

// The mask is rewritten while the kernel runs (lines are erased) and read by other waves of the same
// workgroup: read it past the vector L1.
__device__ __forceinline__ bool mask_on(const uint8_t* m, size_t i) {
    return *reinterpret_cast<const volatile uint8_t*>(m + i) != 0;
}

__device__ __noinline__ void houghp_generic_body(uint8_t* __restrict__ masked, int h, int w, int numrho,
                                                     HoughCfg cfg, unsigned* __restrict__ nz_all,
                                                     const int* __restrict__ npts, int* __restrict__ accum_all,
                                                     const float* __restrict__ trig, int* __restrict__ segs,
                                                     int* __restrict__ nseg, const int* __restrict__ fallback,
                                                     int* __restrict__ path, int rebuild_mask) {
    __shared__ int w_key[3];
    __shared__ int sh_pt[2];
    __shared__ unsigned long long flags[64];           // mask bits of 4096 walk steps
    __shared__ int ends[2][3];                          // x, y, step index of the line end
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (fallback && fallback[s] == 0) return;            // handled by houghp_fast
    if (tid == 0) path[s] = 3;                           // (which kernel made the frame's segments: read by the tests)
    uint8_t* mask = masked + (size_t)s * h * w;
    unsigned* nz = nz_all + (size_t)s * h * w;
    int* accum = accum_all + (size_t)s * NUMANGLE * numrho;
    for (size_t q = tid; q < (size_t)NUMANGLE * numrho; q += 192) accum[q] = 0;
    if (rebuild_mask) {
        // the bit-map resolve pass writes no masked byte map (only this kernel reads one): made here from the point list, after
        // clearing what an earlier frame's walk left behind (w % 16 == 0 on that path)
        for (size_t q = (size_t)tid * 16; q < (size_t)h * w; q += 192 * 16) *reinterpret_cast<uint4*>(mask + q) = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        const int total = npts[s];
        for (int q = tid; q < total; q += 192) mask[(size_t)(nz[q] >> 16) * w + (nz[q] & 0xffffu)] = 255;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const bool th_on = tid < NUMANGLE;
    const float ct = th_on ? trig[2 * tid] : 0.f, sn = th_on ? trig[2 * tid + 1] : 0.f;
    int* arow = accum + (size_t)(th_on ? tid : 0) * numrho + (numrho - 1) / 2;
    unsigned long long rng = ~0ull;
    int nlines = 0;
    const int shift = 16;
    for (int count = npts[s]; count > 0; --count) {
        if (tid == 0) {
            // every thread could do this redundantly; one lane keeps the list traffic minimal
        }
        rng = (unsigned long long)(unsigned)rng * 4164903690ull + (unsigned)(rng >> 32);
        const int idx = (int)((unsigned)rng % (unsigned)count);
        if (tid == 0) {
            const unsigned p = nz[idx];
            nz[idx] = nz[count - 1];
            sh_pt[0] = (int)(p & 0xffffu), sh_pt[1] = (int)(p >> 16);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __syncthreads();
        const int j = sh_pt[0], i = sh_pt[1];
        const bool alive = mask_on(mask, (size_t)i * w + j);
        __syncthreads();
        if (!alive) continue;
        // vote: lane = theta
        int key = -1;
        if (th_on) {
            const int r = __float2int_rn((float)j * ct + (float)i * sn);
            const int val = atomicAdd(&arow[r], 1) + 1;
            key = (val << 8) | (255 - tid);               // max value, then the smallest theta
        }
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
        }
        if (lane == 0) w_key[wid] = key;
        __syncthreads();
        int best = w_key[0] > w_key[1] ? w_key[0] : w_key[1];
        best = best > w_key[2] ? best : w_key[2];
        __syncthreads();
        const int max_val = best >> 8, max_n = 255 - (best & 255);
        if (max_val < cfg.threshold) continue;

        // walk along the line in both directions
        const float a = -trig[2 * max_n + 1], b = trig[2 * max_n];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = __float2int_rn(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        if (tid < 6) ends[tid / 3][tid % 3] = 0;
        __syncthreads();
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            // steps are examined 192 at a time; one lane then replays the gap logic over the flag bits
            int gap = 0, ex = 0, ey = 0, et = 0;
            bool have = false, done = false;
            for (int t0 = 0; !done; t0 += 192) {
                const int t = t0 + tid;
                const int x = x0 + t * dx, y = y0 + t * dy;
                int i1, j1;
                if (xflag) j1 = x, i1 = y >> shift; else j1 = x >> shift, i1 = y;
                const bool inb = j1 >= 0 && j1 < w && i1 >= 0 && i1 < h;
                const bool on = inb && mask_on(mask, (size_t)i1 * w + j1);
                const unsigned long long bon = __ballot(on), bin = __ballot(inb);
                if (lane == 0) flags[wid] = bon, flags[4 + wid] = bin;
                __syncthreads();
                // uniform replay (every thread runs it; cheap and keeps control flow convergent)
                for (int q = 0; q < 192 && !done; ++q) {
                    const unsigned long long fo = flags[q >> 6], fi = flags[4 + (q >> 6)];
                    if (!((fi >> (q & 63)) & 1ull)) { done = true; break; }
                    if ((fo >> (q & 63)) & 1ull) {
                        gap = 0;
                        const int tt = t0 + q, xx = x0 + tt * dx, yy = y0 + tt * dy;
                        if (xflag) ex = xx, ey = yy >> shift; else ex = xx >> shift, ey = yy;
                        et = tt;
                        have = true;
                    } else if (++gap > cfg.line_gap) { done = true; break; }
                }
                __syncthreads();
            }
            if (tid == 0 && have) ends[k][0] = ex, ends[k][1] = ey, ends[k][2] = et;
        }
        __syncthreads();
        const int e0x = ends[0][0], e0y = ends[0][1], e1x = ends[1][0], e1y = ends[1][1];
        const bool good = abs(e1x - e0x) >= cfg.line_length || abs(e1y - e0y) >= cfg.line_length;
        // erase the line's pixels from the mask (start .. line end, both directions); un-vote them if the
        // line is kept.  192 steps per round: every lane clears its own pixel, then all theta lanes
        // subtract the votes of each pixel that was on (atomics: no ordering needed between pixels).
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            const int tend = ends[k][2];
            for (int t0 = 0; t0 <= tend; t0 += 192) {
                const int t = t0 + tid;
                const int x = x0 + t * dx, y = y0 + t * dy;
                int i1, j1;
                if (xflag) j1 = x, i1 = y >> shift; else j1 = x >> shift, i1 = y;
                const bool on = t <= tend && mask_on(mask, (size_t)i1 * w + j1);
                if (on) mask[(size_t)i1 * w + j1] = 0;
                const unsigned long long bon = __ballot(on);
                if (lane == 0) flags[wid] = bon;
                __builtin_amdgcn_s_waitcnt(0x0F70);          // the cleared bytes are out before anyone re-reads
                __syncthreads();
                if (good && th_on) {
                    for (int wq = 0; wq < 3; ++wq) {
                        unsigned long long bits = flags[wq];
                        while (bits) {
                            const int q = __ffsll((long long)bits) - 1;
                            bits &= bits - 1;
                            const int tt = t0 + wq * 64 + q, xx = x0 + tt * dx, yy = y0 + tt * dy;
                            int ii, jj;
                            if (xflag) jj = xx, ii = yy >> shift; else jj = xx >> shift, ii = yy;
                            atomicSub(&arow[__float2int_rn((float)jj * ct + (float)ii * sn)], 1);
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (good) {
            if (tid == 0) {
                int* o = segs + ((size_t)s * cfg.max_segments + nlines) * 4;
                o[0] = e0x, o[1] = e0y, o[2] = e1x, o[3] = e1y;
            }
            if (++nlines >= cfg.max_segments) break;
        }
    }
    if (tid == 0) nseg[s] = nlines;
}

// Fast PPHT: same algorithm and visiting order, restructured for latency.
//   * the point list and a bitmap of the live edge pixels (bounding rows of the points) sit in LDS, so the
//     swap-remove draws, the liveness test and the line walks never touch global memory;
//   * points are voted in speculative batches: up to HB live points are drawn, every theta lane issues the
//     HB returning atomics back to back (one L2 round trip per batch instead of per point).  The returned
//     counts are exactly the sequential ones as long as no point of the batch completes a line; at the
//     first point that does, the later points' votes are rolled back and they are re-examined (liveness
//     included) after the line has been erased, which is what the sequential loop would have seen.
// Streams whose points or bitmap do not fit are flagged and handled by houghp_kernel.
constexpr int HB = 32, NZCAP = 12288, BMWORDS = 16384, FIFO = 128;

__device__ __noinline__ void houghp_fast_body(int h, int w, int numrho, HoughCfg cfg,
                                                   const unsigned* __restrict__ nz_all, const int* __restrict__ npts,
                                                   int* __restrict__ accum_all, const float* __restrict__ trig,
                                                   int* __restrict__ segs, int* __restrict__ nseg,
                                                   int* __restrict__ fallback, int check_flag, int* __restrict__ path) {
    __shared__ unsigned nz[NZCAP];
    __shared__ unsigned bm[BMWORDS];
    __shared__ int fifo[FIFO];             // points drawn from the list but not voted yet (ring buffer)
    __shared__ int bpt[HB];
    __shared__ int didx[HB];
    __shared__ int sh_nb, sh_head, sh_tail, sh_count;
    __shared__ unsigned w_hit[3];
    __shared__ int w_key[3];
    __shared__ unsigned long long flags[8];
    __shared__ int ends[2][3];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned* nzg = nz_all + (size_t)s * h * w;
    int* accum = accum_all + (size_t)s * NUMANGLE * numrho;
    // the flag is both this kernel's input (1: houghp_shard gave the frame up) and its verdict for houghp_kernel, written
    // by thread 0 below: every thread must branch on the value read BEFORE that write, so it goes through LDS
    __shared__ int sh_skip;
    if (tid == 0) sh_skip = (check_flag && fallback[s] == 0) ? 1 : 0;
    __syncthreads();
    if (sh_skip) return;                                        // already done by houghp_shard
    const int total = npts[s];
    const int ymin = total > 0 ? (int)(nzg[0] >> 16) : 0, ymax = total > 0 ? (int)(nzg[total - 1] >> 16) : 0;
    const int wpr = (w + 31) >> 5;
    if (total > NZCAP || (ymax - ymin + 1) * wpr > BMWORDS || cfg.line_gap < 1) {
        if (tid == 0) fallback[s] = 1;
        return;
    }
    if (tid == 0) fallback[s] = 0, path[s] = 2, sh_head = 0, sh_tail = 0, sh_count = total;
    for (int i = tid; i < (ymax - ymin + 1) * wpr; i += 192) bm[i] = 0;
    __syncthreads();
    for (int i = tid; i < total; i += 192) {
        const unsigned p = nzg[i];
        nz[i] = p;
        const int x = (int)(p & 0xffffu), y = (int)(p >> 16);
        atomicOr(&bm[(y - ymin) * wpr + (x >> 5)], 1u << (x & 31));
    }
    __syncthreads();
    auto live = [&](int x, int y) -> bool {
        return y >= ymin && y <= ymax && x >= 0 && x < w && ((bm[(y - ymin) * wpr + (x >> 5)] >> (x & 31)) & 1u);
    };
    const bool th_on = tid < NUMANGLE;
    const float ct = th_on ? trig[2 * tid] : 0.f, sn = th_on ? trig[2 * tid + 1] : 0.f;
    int* arow = accum + (size_t)(th_on ? tid : 0) * numrho + (numrho - 1) / 2;
    // Zero the part of the accumulator this frame can touch (rho range of the points' bounding box per
    // theta, +-2 bins for float rounding).  Doing it here instead of a device-wide memset also leaves the
    // lines in this XCD's L2, where the vote atomics execute.
    {
        int xmn = w, xmx = 0;
        for (int i = tid; i < total; i += 192) {
            const int x = (int)(nz[i] & 0xffffu);
            xmn = x < xmn ? x : xmn, xmx = x > xmx ? x : xmx;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const int a = __shfl_xor(xmn, off, 64), b = __shfl_xor(xmx, off, 64);
            xmn = a < xmn ? a : xmn, xmx = b > xmx ? b : xmx;
        }
        if (lane == 0) w_key[wid] = xmn, w_hit[wid] = (unsigned)xmx;
        __syncthreads();
        xmn = min(w_key[0], min(w_key[1], w_key[2]));
        xmx = (int)max(w_hit[0], max(w_hit[1], w_hit[2]));
        __syncthreads();
        const int half = (numrho - 1) / 2;
        for (int n = 0; n < NUMANGLE; ++n) {
            const float c = trig[2 * n], sv = trig[2 * n + 1];
            const float r0 = (float)xmn * c + (float)ymin * sv, r1 = (float)xmn * c + (float)ymax * sv;
            const float r2 = (float)xmx * c + (float)ymin * sv, r3 = (float)xmx * c + (float)ymax * sv;
            int lo = __float2int_rn(fminf(fminf(r0, r1), fminf(r2, r3))) - 2;
            int hi = __float2int_rn(fmaxf(fmaxf(r0, r1), fmaxf(r2, r3))) + 2;
            lo = lo < -half ? -half : lo, hi = hi > half ? half : hi;
            int* row = accum + (size_t)n * numrho + half;
            for (int r = lo + tid; r <= hi; r += 192) row[r] = 0;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    unsigned long long rng = ~0ull;        // advanced identically by every lane of wave 0
    int nlines = 0;
    const int shift = 16;
    // wave 0: draw up to HB more points from the list into the FIFO (cv::RNG order, swap-remove).
    // The generator is stepped by all lanes (cheap), the modulo is taken lane-parallel, only the
    // swap-remove chain itself is sequential.
    auto top_up = [&]() {
        const int cnt = sh_count, fill = sh_tail - sh_head;
        int nd = FIFO - HB - fill;                // keep room for a rolled-back batch tail (< HB entries)
        nd = nd < HB ? nd : HB;
        nd = nd < cnt ? nd : cnt;
        if (nd <= 0) return;
        unsigned r_mine = 0;
        for (int k = 0; k < nd; ++k) {
            rng = (unsigned long long)(unsigned)rng * 4164903690ull + (unsigned)(rng >> 32);
            if (k == lane) r_mine = (unsigned)rng;
        }
        if (lane < nd) didx[lane] = (int)(r_mine % (unsigned)(cnt - lane));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        if (lane == 0) {
            const int tail = sh_tail;
            for (int k = 0; k < nd; ++k) {
                const int idx = didx[k];
                fifo[(tail + k) & (FIFO - 1)] = (int)nz[idx];
                nz[idx] = nz[cnt - 1 - k];
            }
            sh_tail = tail + nd, sh_count = cnt - nd;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    if (wid == 0) top_up();
    __syncthreads();
    for (;;) {
        // ---- form the batch: pop up to HB points that are still live (wave 0) -------------------------------------
        if (wid == 0) {
            int head = sh_head;
            int nb = 0;
            for (;;) {
                const int avail = sh_tail - head;
                if (avail == 0) {
                    if (sh_count == 0) break;
                    sh_head = head;                       // publish before top_up reads it
                    __builtin_amdgcn_wave_barrier();
                    top_up();
                    continue;
                }
                const int take = avail < (HB - nb) ? avail : (HB - nb);     // <= 32 <= wave
                int p = 0;
                bool ok = false;
                if (lane < take) {
                    p = fifo[(head + lane) & (FIFO - 1)];
                    ok = live(p & 0xffff, p >> 16);
                }
                const unsigned long long okb = __ballot(ok);
                if (ok) bpt[nb + __popcll(okb & ((1ull << lane) - 1ull))] = p;
                nb += __popcll(okb);
                head += take;
                if (nb == HB) break;
            }
            if (lane == 0) sh_head = head, sh_nb = nb;
        }
        __syncthreads();
        const int nb = sh_nb;
        if (nb == 0) break;                 // list and FIFO exhausted
        // ---- speculative votes; wave 0 refills the FIFO while the atomics are in flight ----------------------------
        int val[HB];
        unsigned hitbits = 0;
        // branch-free so the HB returning atomics issue back to back (a predicated block per vote makes the
        // compiler wait for each result in turn): slots past nb and lanes past theta 179 add 0.
        int pts[HB];
#pragma unroll
        for (int b = 0; b < HB; ++b) pts[b] = bpt[b < nb ? b : 0];
#pragma unroll
        for (int b = 0; b < HB; ++b) {
            const int r = __float2int_rn((float)(pts[b] & 0xffff) * ct + (float)(pts[b] >> 16) * sn);
            val[b] = atomicAdd(&arow[r], (b < nb && th_on) ? 1 : 0);
        }
        if (wid == 0) top_up();
#pragma unroll
        for (int b = 0; b < HB; ++b) val[b] = (b < nb && th_on) ? val[b] + 1 : 0;
#pragma unroll
        for (int b = 0; b < HB; ++b)
            if (__ballot(b < nb && th_on && val[b] >= cfg.threshold) != 0ull) hitbits |= 1u << b;
        if (lane == 0) w_hit[wid] = hitbits;
        __syncthreads();
        const unsigned hits = w_hit[0] | w_hit[1] | w_hit[2];
        if (hits == 0) {
            __syncthreads();
            continue;
        }
        const int bs = __ffs((int)hits) - 1;
        int kv = 0;
#pragma unroll
        for (int b = 0; b < HB; ++b)
            if (b == bs) kv = val[b];
        int key = th_on ? ((kv << 8) | (255 - tid)) : -1;
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(key, off, 64);
            key = o > key ? o : key;
        }
        if (lane == 0) w_key[wid] = key;
        // roll back the votes of the points after bs; they are re-examined after the line is erased
#pragma unroll
        for (int b = 0; b < HB; ++b)
            if (b > bs && b < nb && th_on) {
                const int p = bpt[b];
                atomicSub(&arow[__float2int_rn((float)(p & 0xffff) * ct + (float)(p >> 16) * sn)], 1);
            }
        __syncthreads();
        if (tid == 0) {
            // the points after bs go back to the FRONT of the FIFO (they were drawn before everything in it)
            int head = sh_head;
            for (int b = nb - 1; b > bs; --b) fifo[(--head) & (FIFO - 1)] = bpt[b];
            sh_head = head;
        }
        int best = w_key[0] > w_key[1] ? w_key[0] : w_key[1];
        best = best > w_key[2] ? best : w_key[2];
        const int max_n = 255 - (best & 255);
        const int j = bpt[bs] & 0xffff, i = bpt[bs] >> 16;
        // ---- walk along the line in both directions (LDS bitmap) ------------------------------------------------
        const float a = -trig[2 * max_n + 1], b = trig[2 * max_n];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = __float2int_rn(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        if (tid < 6) ends[tid / 3][tid % 3] = 0;
        __syncthreads();
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            int gap = 0, et = 0;
            bool done = false;
            for (int t0 = 0; !done; t0 += 192) {
                const int t = t0 + tid;
                const int x = x0 + t * dx, y = y0 + t * dy;
                int i1, j1;
                if (xflag) j1 = x, i1 = y >> shift; else j1 = x >> shift, i1 = y;
                const bool inb = j1 >= 0 && j1 < w && i1 >= 0 && i1 < h;
                const bool on = inb && live(j1, i1);
                const unsigned long long bon = __ballot(on), bin = __ballot(inb);
                if (lane == 0) flags[wid] = bon, flags[4 + wid] = bin;
                __syncthreads();
                // replay the gap rule on the 192 flag bits, word by word (line_gap >= 1)
                for (int wq = 0; wq < 3 && !done; ++wq) {
                    unsigned long long fo = flags[wq];
                    const unsigned long long fi = flags[4 + wq];
                    const int nin = fi == ~0ull ? 64 : __ffsll((long long)~fi) - 1;      // in-bounds steps of this word
                    int pos = 0;                                                          // steps consumed
                    while (pos < nin) {
                        const unsigned long long rest = fo >> pos;
                        if (rest == 0ull) {                                               // only gaps remain
                            const int zeros = nin - pos;
                            if (gap + zeros > cfg.line_gap) done = true;
                            gap += zeros;
                            pos = nin;
                            break;
                        }
                        const int z = __ffsll((long long)rest) - 1;                       // zeros before the next hit
                        if (z >= nin - pos) {
                            const int zeros = nin - pos;
                            if (gap + zeros > cfg.line_gap) done = true;
                            gap += zeros;
                            pos = nin;
                            break;
                        }
                        if (gap + z > cfg.line_gap) { done = true; break; }
                        pos += z;
                        gap = 0;
                        et = t0 + wq * 64 + pos;
                        ++pos;
                    }
                    if (nin < 64) done = true;                                            // left the image
                }
                __syncthreads();
            }
            if (tid == 0) {
                const int xx = x0 + et * dx, yy = y0 + et * dy;
                ends[k][0] = xflag ? xx : xx >> shift, ends[k][1] = xflag ? yy >> shift : yy, ends[k][2] = et;
            }
        }
        __syncthreads();
        const int e0x = ends[0][0], e0y = ends[0][1], e1x = ends[1][0], e1y = ends[1][1];
        const bool good = abs(e1x - e0x) >= cfg.line_length || abs(e1y - e0y) >= cfg.line_length;
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            const int tend = ends[k][2];
            for (int t0 = 0; t0 <= tend; t0 += 192) {
                const int t = t0 + tid;
                const int x = x0 + t * dx, y = y0 + t * dy;
                int i1, j1;
                if (xflag) j1 = x, i1 = y >> shift; else j1 = x >> shift, i1 = y;
                const bool on = t <= tend && live(j1, i1);
                const unsigned long long bon = __ballot(on);
                if (lane == 0) flags[wid] = bon;
                __syncthreads();
                if (on) atomicAnd(&bm[(i1 - ymin) * wpr + (j1 >> 5)], ~(1u << (j1 & 31)));
                if (good && th_on) {
                    for (int wq = 0; wq < 3; ++wq) {
                        unsigned long long bits = flags[wq];
                        while (bits) {
                            const int q = __ffsll((long long)bits) - 1;
                            bits &= bits - 1;
                            const int tt = t0 + wq * 64 + q, xx = x0 + tt * dx, yy = y0 + tt * dy;
                            int ii, jj;
                            if (xflag) jj = xx, ii = yy >> shift; else jj = xx >> shift, ii = yy;
                            atomicSub(&arow[__float2int_rn((float)jj * ct + (float)ii * sn)], 1);
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (good) {
            if (tid == 0) {
                int* o = segs + ((size_t)s * cfg.max_segments + nlines) * 4;
                o[0] = e0x, o[1] = e0y, o[2] = e1x, o[3] = e1y;
            }
            if (++nlines >= cfg.max_segments) break;
        }
        __syncthreads();
    }
    if (tid == 0) nseg[s] = nlines;
}

// ---- L4b': PPHT with theta sharded over workgroups AND waves, accumulator in LDS ---------------------------------------
// houghp_fast is bound by its CU's L1/TA pipeline: every theta lane votes into its own 16-KB row of a global
// accumulator, one cache line per lane per vote.  Here a frame is handled by HG = 4 workgroups of HVW = 4 voting waves (+ a helper
// wave): voting wave v of workgroup g is sub-shard q = 4 g + v of 16 and owns theta = 16 tl + q, tl = 0 .. 11 -- twelve rows, kept in
// LDS over the rho range the points' bounding box can reach (variable-length rows), as 16-bit counters biased by 0x4000 and packed
// two to a word (PPHT's erase also decrements pixels that have not voted yet, so counts go negative; |count| <= number of points
// <= 4096).  A workgroup keeps ONE copy of the point list, RNG position, FIFO and live-pixel bitmap and replays the sequential
// part of the algorithm once (wave 0: batch forming, line walk; helper wave: FIFO top-ups, the second direction of every erase);
// the four workgroups of a frame replay it identically -- all deterministic -- so they stay in step by construction and exchange
// just two words per batch through global memory: the mask of batch points whose vote reached the threshold, and, when a line
// fires, their best (count, theta) key.  A word carries its batch sequence number and is stored/loaded with agent scope (the XCDs'
// L2s are not coherent); slots are double-buffered by sequence parity; every spin is bounded and a timeout hands the frame to
// houghp_fast.  Inside a workgroup the waves meet at workgroup barriers (phases below) and pass masks / keys through LDS.
//
// Round 4: the votes.  Round 3 had one voting wave per workgroup, lane = theta row (45 of 64 lanes), one returning LDS atomic per
// batch point: 32 dependent-issue instructions per batch on a lone wave, 47 % of the kernel.  Now a voting lane is (theta row tl,
// point slot sub) -- 12 x 5 = 60 lanes -- and ONE returning atomic instruction casts the votes of five consecutive batch points on
// twelve rows: a batch of 32 is 7 instructions per wave, on four waves side by side.  A vote's returned count must include exactly
// the batch's EARLIER points on the same cell; the points of one instruction that hit the same cell sit in lanes tl, tl + 12,
// tl + 24, ... in batch order, and the LDS serves the lanes of one atomic instruction that address the same word in increasing lane
// order, each with the running value (tools/wldsorder.hip: 0 violations in 4 x 65 536 lanes over all-same, paired, quadruple and
// random address patterns, with and without a second wave hammering the same banks); rows of different waves are disjoint, so
// the waves need no order among themselves.  (Round 3 had tried two waves voting halves of a batch on the SAME rows: one run in
// five lost or gained a segment.)  Sixteen single-wave shards in sixteen workgroups of 79 KB -- two per CU, or one beside a
// convolution workgroup -- were built first: same segments, but 1 024 workgroups are two rounds of residency on 256 CUs, 422 us
// per 64 frames against 277; the replicated point list + bitmap (52 KB) are what keeps a finer split from fitting.
constexpr int HG = 4, HVW = 4, HQ = HG * HVW, HTL = 12, HNS = 5, HNV = (HB + HNS - 1) / HNS;    // workgroups per frame, voting waves, sub-shards,
                                                                  // theta rows per sub-shard, point slots per instruction, vote instructions per batch
constexpr int HS_NZ = 4096, HS_BMW = 9216, HS_ACCW = 6144;        // LDS capacities: points, bitmap words (288 rows x 32 words), accumulator words per voting wave
constexpr int HS_EB = 48;                                         // 64-step words of one direction of a line walk (3 072 steps)
constexpr unsigned HS_BIAS = 0x4000u;
constexpr int HS_SPIN = 1 << 20;              // default bound of every exchange spin (x s_sleep 2), ~0.1 s; AVHOT_HOUGH_SPIN overrides it
static_assert(HQ * HTL >= NUMANGLE && HTL * HNS <= 64 && HG <= 64, "shard geometry");

__global__ void hough_prep_kernel(int n_streams, int numrho, int* __restrict__ accum_all, int* __restrict__ fallback) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams * 8 * HG) return;
    const int s = i / (8 * HG), k = i - s * (8 * HG);
    accum_all[(size_t)s * NUMANGLE * numrho + k] = 0;              // 2 rounds x 2 parities x HG exchange words (8 bytes each)
    if (k == 0) fallback[s] = 0;
}

// cv::RNG(-1) is created afresh by every HoughLinesP call, so the draws of a frame are always the same sequence:
// a table built at compile time replaces the multiply-with-carry chain (a 64-bit multiply per draw, serial).
struct HoughDraws { unsigned v[HS_NZ]; };
constexpr HoughDraws hough_draws() {
    HoughDraws t{};
    unsigned long long st = ~0ull;
    for (int k = 0; k < HS_NZ; ++k) {
        st = (unsigned long long)(unsigned)st * 4164903690ull + (unsigned)(st >> 32);
        t.v[k] = (unsigned)st;
    }
    return t;
}
__device__ const HoughDraws g_draws = hough_draws();

__global__ void __launch_bounds__((HVW + 1) * 64) houghp_shard(int h, int w, int numrho, HoughCfg cfg, const unsigned* __restrict__ nz_all,
                                                   const int* __restrict__ npts, int* __restrict__ accum_all,
                                                   const float* __restrict__ trig, int* __restrict__ segs,
                                                   int* __restrict__ nseg, int* __restrict__ fallback, int spin_limit,
                                                   int drop_frame, int* __restrict__ path, int timed, int xcd_local) {
    // spin_limit: iterations an exchange waits for a partner before the frame is handed to houghp_fast; drop_frame (tests only,
    // AVHOT_HOUGH_DROP): workgroup HG-1 of that frame never publishes its first exchange word, so its partners run into the limit;
    // timed (AVHOT_HOUGH_TIMED, tools/htime.py): wave 0 of workgroup 0 adds up s_memtime cycles per phase and leaves them behind
    // the frame's exchange words (accumulator view, 64-bit words 32 .. 47)
    __shared__ unsigned acc_all[HVW][HS_ACCW];
    __shared__ unsigned nz[HS_NZ];
    __shared__ unsigned bm[HS_BMW];
    __shared__ int fifo[FIFO];
    __shared__ int bpt[HB];
    __shared__ float bfx[HB], bfy[HB];
    __shared__ int row_sz[HQ * HTL];
    __shared__ unsigned long long ebits[2][HS_EB];             // live pixels found (and cleared) by the two directions of an erase
    __shared__ unsigned tu_cnt[256];                           // top-up: how many of a top-up's draws fall on a position class (idx & 255)
    __shared__ unsigned sh_hit[HVW], sh_key[HVW];
    __shared__ int sh_nb, sh_head, sh_tail, sh_count, sh_big, sh_hits, sh_fail, sh_er[8];
    // Workgroups go round-robin over the 8 XCDs (blockIdx % 8): the four workgroups of a frame are given the same blockIdx % 8, so that
    // their exchange words meet in one XCD's L2 (AVHOT_HOUGH_XCD=0: consecutive workgroups per frame).  Placement only: any map is correct.
    int s = blockIdx.x / HG, g = blockIdx.x % HG;
    if (xcd_local && (gridDim.x % (8 * HG)) == 0) {
        const int b = blockIdx.x, blk = b / (8 * HG), r = b - blk * (8 * HG);
        s = blk * 8 + (r & 7), g = r >> 3;
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);       // 0 .. HVW-1: voting waves (0 also runs the sequential part); HVW: helper
    const bool voter = wv < HVW;
    const unsigned* nzg = nz_all + (size_t)s * h * w;
    unsigned long long* xw = reinterpret_cast<unsigned long long*>(accum_all + (size_t)s * NUMANGLE * numrho);
    const int total = npts[s];
    auto give_up = [&]() {
        if (wv == 0 && lane == 0) fallback[s] = 1;
    };
    auto lds_order = [&]() {                       // one wave: order its LDS writes before later reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    };
    constexpr int NTH = (HVW + 1) * 64;
    unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = timed ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](int k) {                        // cycles since the previous lap -> phase k
        if (timed) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            tph[k] += t - tlast, tlast = t;
        }
    };
    // the LDS clears do not depend on the point list: they run while its loads are in flight
    for (int i = (int)threadIdx.x; i < HVW * HS_ACCW; i += NTH) (&acc_all[0][0])[i] = HS_BIAS | (HS_BIAS << 16);
    for (int i = (int)threadIdx.x; i < HS_BMW; i += NTH) bm[i] = 0;
    for (int i = (int)threadIdx.x; i < 256; i += NTH) tu_cnt[i] = 0;
    if (total > HS_NZ) {                                                                  // same verdict in all HG workgroups (and every wave)
        give_up();
        return;
    }
    // ---- set-up: bounding box of the points (every wave computes it: same loads, same verdicts), list + bitmap into LDS -------
    constexpr int NZ_PER = HS_NZ / 64;
    int ymax = 0, xmn = w, xmx = 0;                                                       // the list is sorted by (y, x)
    int ymin = 0;
    {
        unsigned first = 0;
#pragma unroll 8
        for (int k = 0; k < NZ_PER; ++k) {
            const bool on = k * 64 + lane < total;
            const unsigned p = on ? nzg[k * 64 + lane] : 0u;
            if (k == 0) first = p;
            const int x = (int)(p & 0xffffu), y = (int)(p >> 16);
            ymax = on && y > ymax ? y : ymax;
            xmn = on && x < xmn ? x : xmn, xmx = on && x > xmx ? x : xmx;
        }
        ymin = total > 0 ? (int)((unsigned)__builtin_amdgcn_readfirstlane((int)first) >> 16) : 0;
    }
    ymax = (int)wave_max_u32((unsigned)ymax);
    xmx = (int)wave_max_u32((unsigned)xmx);
    xmn = w - (int)wave_max_u32((unsigned)(w - xmn));
    if (total == 0) xmn = 0, xmx = 0;
    // the live-pixel bitmap covers the points' rows and the 32-bit words of their column range only
    const int wlo = xmn >> 5, wpr = (xmx >> 5) - wlo + 1, xlo = wlo << 5, xhi = xlo + (wpr << 5);
    if ((ymax - ymin + 1) * wpr > HS_BMW || cfg.line_gap < 1) {
        give_up();
        return;
    }
    __syncthreads();                               // the clears are done before anybody fills the bitmap
    for (int i = (int)threadIdx.x; i < total; i += NTH) {
        const unsigned p = nzg[i];
        nz[i] = p;
        const int x = (int)(p & 0xffffu), y = (int)(p >> 16);
        atomicOr(&bm[(y - ymin) * wpr + ((x - xlo) >> 5)], 1u << (x & 31));
    }
    if (threadIdx.x == 0) sh_head = 0, sh_tail = 0, sh_count = total, sh_fail = 0;
    auto bm_word = [&](int x, int y) -> unsigned* { return &bm[(y - ymin) * wpr + ((x - xlo) >> 5)]; };
    auto live = [&](int x, int y) -> bool {
        return y >= ymin && y <= ymax && x >= xlo && x < xhi && ((*bm_word(x, y) >> (x & 31)) & 1u);
    };
    // voting lane = (theta row tl, point slot sub): lanes 0 .. 59; theta = HQ * tl + q, q = HVW * g + wave
    const int sub = lane / HTL, tl = lane - sub * HTL;
    const int q_mine = g * HVW + (voter ? wv : 0);
    const int th = tl * HQ + q_mine;
    const bool th_on = voter && sub < HNS && th < NUMANGLE;
    const float ct = th_on ? trig[2 * th] : 0.f, sn = th_on ? trig[2 * th + 1] : 0.f;
    // the accumulator rows: the rho range of the points' bounding box for every theta (+-2 bins).  The capacity verdict has to be
    // the same in all HG workgroups of the frame, so each one sizes every sub-shard's twelve rows.
    const int half = (numrho - 1) / 2;
    auto row_lo_hi = [&](float c2, float s2, int& lo, int& hi) {
        const float r0 = (float)xmn * c2 + (float)ymin * s2, r1 = (float)xmn * c2 + (float)ymax * s2;
        const float r2 = (float)xmx * c2 + (float)ymin * s2, r3 = (float)xmx * c2 + (float)ymax * s2;
        lo = __float2int_rn(fminf(fminf(r0, r1), fminf(r2, r3))) - 2;
        hi = __float2int_rn(fmaxf(fmaxf(r0, r1), fmaxf(r2, r3))) + 2;
        lo = lo < -half ? -half : lo, hi = hi > half ? half : hi;
    };
    if (wv == 0) {
        for (int t2 = lane; t2 < HQ * HTL; t2 += 64) {               // t2 = theta
            int lo = 0, hi = -1;
            if (t2 < NUMANGLE) row_lo_hi(trig[2 * t2], trig[2 * t2 + 1], lo, hi);
            row_sz[t2] = hi - lo + 1;
        }
        lds_order();
        bool too_big = false;
        if (lane < HQ) {                                             // lane q: the cells sub-shard q needs
            int tot = 0;
            for (int r = 0; r < HTL; ++r) tot += row_sz[r * HQ + lane];
            too_big = tot > 2 * HS_ACCW;
        }
        const unsigned long long bigm = __ballot(too_big);
        if (lane == 0) sh_big = bigm != 0ull ? 1 : 0;
    }
    __syncthreads();                               // list, bitmap, sizes and the verdict are in LDS
    if (sh_big) {
        give_up();
        return;
    }
    int my_off = 0;
    if (th_on) {
        int my_base = 0, lo, hi;
        for (int r = 0; r < tl; ++r) my_base += row_sz[r * HQ + q_mine];
        row_lo_hi(ct, sn, lo, hi);
        my_off = my_base - lo;
    }
    unsigned* const acc = acc_all[voter ? wv : 0];
    // cell index of (this lane's theta, rho of the pixel); idle lanes have ct = sn = 0, my_off = 0 and land on cell 0 (they add 0)
    auto cellf = [&](float fx, float fy) { return __float2int_rn(fx * ct + fy * sn) + my_off; };
    auto vote = [&](int B, unsigned add) -> int {          // count BEFORE the vote (add = 0: a plain read)
        const int sh = (B & 1) * 16;
        const unsigned old = atomicAdd(&acc[B >> 1], add << sh);
        return (int)((old >> sh) & 0xFFFFu) - (int)HS_BIAS;
    };
    auto unvote = [&](int B) { atomicSub(&acc[B >> 1], 1u << ((B & 1) * 16)); };
    // one 64-bit word per workgroup, round and sequence parity: (sequence << 32) | payload, agent scope.  Wave 0 only; lane k < HG
    // polls partner k; the payloads are combined by the caller (OR of hit masks / maximum of keys)
    unsigned seq = 0;
    auto exchange = [&](int round, unsigned payload, unsigned& of_partner) -> bool {
        unsigned long long* slot = xw + (size_t)(round * 2 + (seq & 1)) * HG;
        unsigned long long wd = ((unsigned long long)seq << 32) | payload;
        const bool dropped = s == drop_frame && g == HG - 1 && seq == 1u && round == 0;
        if (lane == 0 && !dropped) __hip_atomic_store(slot + g, wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        if (lane < HG && lane != g) {
            int it = 0;
            for (;;) {
                wd = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(wd >> 32) == seq) break;
                if (++it > spin_limit) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (__ballot(!ok) != 0ull) return false;
        of_partner = lane < HG ? (unsigned)wd : 0u;                 // (lane g holds its own payload)
        return true;
    };
    const int shift = 16;
    const int kk = lane & (HB - 1);                 // both half-waves work on draw kk of a top-up
    // FIFO top-up (one wave; normally the helper, beside the votes of the batch just formed).  The draws are a function of how many
    // points were drawn before (cv::RNG(-1) per call: a compile-time table); the helper keeps the next top-up's draws in flight.
    int pre_base = -1;
    unsigned pre_r = 0;
    auto top_up = [&](bool prefetch) {
        const int cnt = sh_count, fill = sh_tail - sh_head;
        int nd = FIFO - HB - fill;
        nd = nd < HB ? nd : HB;
        nd = nd < cnt ? nd : cnt;
        if (nd <= 0) return;
        const int base = total - cnt;
        const unsigned r_mine = pre_base == base ? pre_r : (base + kk < total ? g_draws.v[base + kk] : 0u);
        if (prefetch) {
            pre_base = base + nd;
            pre_r = pre_base + kk < total ? g_draws.v[pre_base + kk] : 0u;
        }
        // cv's draw: step k takes nz[idx_k] and moves nz[cnt-1-k] into its place -- a chain when replayed literally.
        // Resolved in registers instead: what position p holds at step k is what the ORIGINAL array holds at the
        // position found by chasing p backwards through the earlier steps that wrote it.  The low half-wave does
        // that for the picks, the high half for the elements moved in; only the last writer of a position stores.
        // Round 4: a step j can redirect anybody's chase only if idx_j equals another draw's index or lies in the tail region
        // [cnt - nd, cnt) the moved-in elements come from -- otherwise idx_j matches no chased position at all.  Those steps are
        // found first (a 256-entry count table over idx & 255: a count above one flags every draw of a class that holds a
        // duplicate, and a few innocent ones) and the chase visits only them, in the original descending order: typically 3-6
        // steps instead of 32 (round 3's 32-step loop was ~3 000 cycles per top-up, and a frame draws ALL its points).
        const int idx = kk < nd ? (int)(r_mine % (unsigned)(cnt - kk)) : -1;
        const bool drawer = lane < HB && kk < nd;
        if (drawer) atomicAdd(&tu_cnt[idx & 255], 1u);
        lds_order();
        const bool involved = drawer && (tu_cnt[idx & 255] > 1u || idx >= cnt - nd);
        unsigned hot = (unsigned)__ballot(involved);              // (low half-wave: bit = draw index)
        lds_order();
        if (drawer) atomicSub(&tu_cnt[idx & 255], 1u);
        int p = lane < HB ? idx : cnt - 1 - kk;
        bool last_writer = true;
        while (hot) {
            const int k2 = 31 - __clz((int)hot);
            hot &= ~(1u << k2);
            const int ik = __builtin_amdgcn_readlane(idx, k2);
            if (k2 < kk && ik == p) p = cnt - 1 - k2;
            if (k2 > kk && ik == idx) last_writer = false;
        }
        unsigned v = 0;
        if (kk < nd) v = nz[p];
        lds_order();
        if (kk < nd) {
            if (lane < HB) fifo[(sh_tail + kk) & (FIFO - 1)] = (int)v;
            else if (last_writer) nz[idx] = v;
        }
        lds_order();
        if (lane == 0) sh_tail = sh_tail + nd, sh_count = cnt - nd;
        lds_order();
    };
    // ---- the direction-k half of a fired line's erase, bitmap part (one wave): test and clear the live pixels of the walk, 64 steps
    // per trip, and leave the found bits for the voting waves (direction 1 starts behind the fired pixel: direction 0 erases that one)
    auto erase_bitmap = [&](int k) {
        const int x0 = sh_er[0], y0 = sh_er[1], dx0 = sh_er[2], dy0 = sh_er[3], xflag = sh_er[4], tend = sh_er[6 + k];
        const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
        for (int t0 = 0, c = 0; t0 <= tend; t0 += 64, ++c) {
            const int t = t0 + lane;
            const int x = x0 + t * dx, y = y0 + t * dy;
            int i1, j1;
            if (xflag) j1 = x, i1 = y >> 16; else j1 = x >> 16, i1 = y;
            const bool on = t <= tend && t >= k && live(j1, i1);
            const unsigned long long bits = __ballot(on);
            if (on) atomicAnd(bm_word(j1, i1), ~(1u << (j1 & 31)));
            if (lane == 0) ebits[k][c] = bits;
            lds_order();
        }
    };
    // ... and the accumulator part (every voting wave, its own rows): the votes of the pixels found live are taken back five at a
    // time -- point slot `sub` takes the set bits of its own 13-bit slice of a 64-step word (decrements commute: any order)
    auto erase_votes = [&]() {
        const int x0 = sh_er[0], y0 = sh_er[1], dx0 = sh_er[2], dy0 = sh_er[3], xflag = sh_er[4];
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0, tend = sh_er[6 + k];
            for (int t0 = 0, c = 0; t0 <= tend; t0 += 64, ++c) {
                const unsigned long long bits = ebits[k][c];
                if (bits == 0ull) continue;
                unsigned mb = sub < HNS ? (unsigned)(bits >> (13 * sub)) & 0x1FFFu : 0u;
                while (mb) {
                    const int q = 13 * sub + __ffs((int)mb) - 1;
                    mb &= mb - 1;
                    const int tt = t0 + q, xx = x0 + tt * dx, yy = y0 + tt * dy;
                    int ii, jj;
                    if (xflag) jj = xx, ii = yy >> 16; else jj = xx >> 16, ii = yy;
                    if (th_on) unvote(cellf((float)jj, (float)ii));
                }
            }
        }
    };
    int nlines = 0;
    if (wv == HVW) top_up(true);
    __syncthreads();
    lap(0);                                        // set-up
    // Phases of one batch; every wave passes the same barriers (B1 .. B6), all tests between them are workgroup-uniform (LDS words).
    for (;;) {
        // ---- wave 0: form the batch: pop up to HB points that are still live ------------------------------------------------------
        if (wv == 0) {
            int head = sh_head;
            int nb = 0;
            for (;;) {
                const int avail = sh_tail - head;
                if (avail == 0) {
                    if (sh_count == 0) break;
                    if (lane == 0) sh_head = head;
                    lds_order();
                    top_up(false);                 // (rare: the FIFO ran dry inside a batch -- the helper is idle in this phase)
                    continue;
                }
                const int take = avail < (HB - nb) ? avail : (HB - nb);
                int p = 0;
                bool ok = false;
                if (lane < take) {
                    p = fifo[(head + lane) & (FIFO - 1)];
                    ok = live(p & 0xffff, p >> 16);
                }
                const unsigned long long okb = __ballot(ok);
                if (ok) {
                    const int slot = nb + __popcll(okb & ((1ull << lane) - 1ull));
                    bpt[slot] = p, bfx[slot] = (float)(p & 0xffff), bfy[slot] = (float)(p >> 16);
                }
                nb += __popcll(okb);
                head += take;
                if (nb == HB) break;
            }
            if (lane == 0) sh_head = head, sh_nb = nb;
        }
        lap(1);                                    // batch forming
        __syncthreads();                           // B1: the batch is in LDS
        const int nb = sh_nb;
        if (nb == 0) break;
        // ---- voting waves: speculative votes, instruction v = batch points 5 v .. 5 v + 4; helper: the next top-up ---------------
        int val[HNV], cel[HNV];
        if (voter) {
            unsigned hm = 0;                             // per lane: its points whose count reached the threshold here (bit = batch index)
#pragma unroll
            for (int v = 0; v < HNV; ++v) {
                const int b = v * HNS + sub;
                const bool bon = th_on && b < nb;
                cel[v] = bon ? cellf(bfx[b < HB ? b : 0], bfy[b < HB ? b : 0]) : 0;
            }
#pragma unroll
            for (int v = 0; v < HNV; ++v) {
                const int b = v * HNS + sub;
                const bool bon = th_on && b < nb;
                val[v] = vote(cel[v], bon ? 1u : 0u);
                hm |= (bon && val[v] + 1 >= cfg.threshold) ? 1u << (b & 31) : 0u;
            }
            const unsigned hb = wave_or_u32(hm);
            if (lane == 0) sh_hit[wv] = hb;
        } else {
            top_up(true);
        }
        __syncthreads();                           // B2: hit masks of the four voting waves, FIFO refilled
        lap(2);                                    // votes (and the helper's top-up) incl. both barriers
        tph[10] += 1;
        if (wv == 0) {
            unsigned hitbits = 0;
#pragma unroll
            for (int v = 0; v < HVW; ++v) hitbits |= sh_hit[v];
            seq += 1;
            unsigned got = 0;
            const bool ok = exchange(0, hitbits, got);
            const unsigned hits = wave_or_u32(got);
            if (lane == 0) sh_hits = (int)hits, sh_fail = ok ? 0 : 1;
        }
        __syncthreads();                           // B3: the frame's hit mask
        lap(3);                                    // first exchange
        if (sh_fail) {
            give_up();
            return;
        }
        const unsigned hits = (unsigned)sh_hits;
        if (hits == 0) continue;
        const int bs = __ffs((int)hits) - 1;
        if (voter) {
            const int bsv = bs / HNS, bss = bs - bsv * HNS;
            int kv = 0;
#pragma unroll
            for (int v = 0; v < HNV; ++v)
                if (v == bsv) kv = val[v] + 1;
            // counts can be negative (pixels erased before they voted): order-preserving signed -> unsigned map, 0 = no theta here
            const unsigned mykey = (th_on && sub == bss) ? ((unsigned)((kv << 8) | (255 - th)) ^ 0x80000000u) : 0u;
            const unsigned lb = wave_max_u32(mykey);
            if (lane == 0) sh_key[wv] = lb;
            // the votes of the points after bs are withdrawn; they are re-examined after the line is erased
#pragma unroll
            for (int v = 0; v < HNV; ++v) {
                const int b = v * HNS + sub;
                if (b > bs && b < nb && th_on) unvote(cel[v]);
            }
        }
        __syncthreads();                           // B4: the workgroup's best keys
        lap(4);                                    // keys, withdrawn votes
        tph[11] += 1;
        if (wv == 0) {
            unsigned lbest = 0;
#pragma unroll
            for (int v = 0; v < HVW; ++v) lbest = sh_key[v] > lbest ? sh_key[v] : lbest;
            unsigned got = 0;
            const bool ok = exchange(1, lbest, got);
            unsigned best = wave_max_u32(got);
            best ^= 0x80000000u;
            lap(5);                                // second exchange
            if (lane == 0) {
                int head = sh_head;
                for (int b = nb - 1; b > bs; --b) fifo[(--head) & (FIFO - 1)] = bpt[b];
                sh_head = head;
            }
            lds_order();
            const int max_n = 255 - (int)(best & 255u);
            const int j = bpt[bs] & 0xffff, i = bpt[bs] >> 16;
            // ---- walk along the line in both directions ----------------------------------------------------------
            const float a = -trig[2 * max_n + 1], b = trig[2 * max_n];
            int x0 = j, y0 = i, dx0, dy0, xflag;
            if (fabsf(a) > fabsf(b)) {
                xflag = 1;
                dx0 = a > 0 ? 1 : -1;
                dy0 = __float2int_rn(b * (float)(1 << shift) / fabsf(a));
                y0 = (y0 << shift) + (1 << (shift - 1));
            } else {
                xflag = 0;
                dy0 = b > 0 ? 1 : -1;
                dx0 = __float2int_rn(a * (float)(1 << shift) / fabsf(b));
                x0 = (x0 << shift) + (1 << (shift - 1));
            }
            int ends[2][3];
            for (int k = 0; k < 2; ++k) {
                const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
                int gap = 0, et = 0;
                bool done = false;
                for (int t0 = 0; !done; t0 += 64) {
                    const int t = t0 + lane;
                    const int x = x0 + t * dx, y = y0 + t * dy;
                    int i1, j1;
                    if (xflag) j1 = x, i1 = y >> shift; else j1 = x >> shift, i1 = y;
                    const bool inb = j1 >= 0 && j1 < w && i1 >= 0 && i1 < h;
                    const bool on = inb && live(j1, i1);
                    const unsigned long long fo = __ballot(on), fi = __ballot(inb);
                    const int nin = fi == ~0ull ? 64 : __ffsll((long long)~fi) - 1;         // in-bounds steps of this word
                    int pos = 0;
                    if (cfg.line_gap >= 63 && nin > 0) {           // no run of zeros inside one word can exceed the gap
                        const unsigned long long f = nin == 64 ? fo : fo & ((1ull << nin) - 1ull);
                        if (f == 0ull) {
                            if (gap + nin > cfg.line_gap) done = true;
                            gap += nin;
                        } else if (gap + (__ffsll((long long)f) - 1) > cfg.line_gap) {
                            done = true;
                        } else {
                            const int last = 63 - __clzll((long long)f);
                            et = t0 + last;
                            gap = nin - 1 - last;
                        }
                        pos = nin;
                    }
                    while (pos < nin) {                                                        // the gap rule (line_gap >= 1)
                        const unsigned long long rest = fo >> pos;
                        int z = rest == 0ull ? 64 : __ffsll((long long)rest) - 1;            // zeros before the next hit
                        if (z >= nin - pos) {
                            const int zeros = nin - pos;
                            if (gap + zeros > cfg.line_gap) done = true;
                            gap += zeros;
                            pos = nin;
                            break;
                        }
                        if (gap + z > cfg.line_gap) { done = true; break; }
                        pos += z;
                        gap = 0;
                        et = t0 + pos;
                        ++pos;
                    }
                    if (nin < 64) done = true;                                                 // left the image
                }
                const int xx = x0 + et * dx, yy = y0 + et * dy;
                ends[k][0] = xflag ? xx : xx >> shift, ends[k][1] = xflag ? yy >> shift : yy, ends[k][2] = et;
            }
            const int e0x = ends[0][0], e0y = ends[0][1], e1x = ends[1][0], e1y = ends[1][1];
            const bool good = abs(e1x - e0x) >= cfg.line_length || abs(e1y - e0y) >= cfg.line_length;
            const bool fits = ends[0][2] < 64 * HS_EB && ends[1][2] < 64 * HS_EB;              // (walks longer than the bit buffer: houghp_fast)
            if (lane == 0) {
                sh_er[0] = x0, sh_er[1] = y0, sh_er[2] = dx0, sh_er[3] = dy0, sh_er[4] = xflag, sh_er[5] = good ? 1 : 0;
                sh_er[6] = ends[0][2], sh_er[7] = ends[1][2];
                sh_fail = (ok && fits) ? 0 : 1;
                if (good && ok && fits && g == 0) {
                    int* o = segs + ((size_t)s * cfg.max_segments + nlines) * 4;
                    o[0] = e0x, o[1] = e0y, o[2] = e1x, o[3] = e1y;
                }
            }
        }
        __syncthreads();                           // B5: the line
        lap(6);                                    // line walk
        if (sh_fail) {
            give_up();
            return;
        }
        if (wv == 0) erase_bitmap(0);              // this wave direction 0, the helper direction 1
        else if (wv == HVW) erase_bitmap(1);
        __syncthreads();                           // B6: the erased pixels' bits
        lap(7);                                    // erase, bitmap part
        const bool good = sh_er[5] != 0;
        if (good && voter) erase_votes();          // (runs beside wave 0's next batch forming: the rows are this wave's own)
        lap(8);                                    // erase, accumulator part
        if (good && ++nlines >= cfg.max_segments) break;
    }
    if (wv == 0 && g == 0 && lane == 0) nseg[s] = nlines, path[s] = 1;
    if (timed && wv == 0 && g == 0 && lane == 0)
        for (int k = 0; k < 12; ++k) xw[32 + k] = tph[k];
}

// ---- L5-L7: slope split, quadratic fit, EMA, resampling ------------------------------------------------------
// one Jacobi rotation in the (P, Q) plane; P, Q compile-time so that A and V stay in registers (run-time indices put them
// in scratch memory, which made this 3x3 problem a 40-us kernel)
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&A)[3][3], double (&V)[3][3]) {
    if (fabs(A[P][Q]) < 1e-300) return;
    const double th = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);
    const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double akp = A[k][P], akq = A[k][Q];
        A[k][P] = c * akp - sn * akq, A[k][Q] = sn * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double apk = A[P][k], aqk = A[Q][k];
        A[P][k] = c * apk - sn * aqk, A[Q][k] = sn * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[k][P], vkq = V[k][Q];
        V[k][P] = c * vkp - sn * vkq, V[k][Q] = sn * vkp + c * vkq;
    }
}
__device__ __forceinline__ void jacobi3(double (&A)[3][3], double (&V)[3][3]) {        // symmetric eigen-decomposition, A -> diag
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) V[r][c] = r == c ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        // converged when the off-diagonal mass is below 1e-20 of the diagonal's: four orders of magnitude under double
        // rounding, reached after 4-6 sweeps (a fixed 1e-300 needed ~10, each three dependent divide + square-root chains)
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= 1e-20 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2])) || off < 1e-300) break;
        jacobi_rot<0, 1>(A, V);
        jacobi_rot<0, 2>(A, V);
        jacobi_rot<1, 2>(A, V);
    }
}

__device__ void lane_fit_one(int s, int side, int h, int w, int max_segments, double smoothing, const int* __restrict__ segs,
                             const int* __restrict__ nseg, double* __restrict__ lane_state, double* __restrict__ poly,
                             int* __restrict__ info, double* __restrict__ conf, const double* __restrict__ thr,
                             const int* __restrict__ npts, double* coef) {
    const int n = nseg[s];
    const int* sg = segs + (size_t)s * max_segments * 4;
    const double cx = (double)w / 2.0;
    // lstsq on the Vandermonde matrix [y^2 y 1] with numpy.polyfit's column scaling
    double s4 = 0, s3 = 0, s2 = 0, s1 = 0, s0 = 0, t2 = 0, t1 = 0, t0 = 0;
    int nl = 0;
    for (int k = 0; k < n; ++k) {
        const int x1 = sg[4 * k], y1 = sg[4 * k + 1], x2 = sg[4 * k + 2], y2 = sg[4 * k + 3];
        if (x2 == x1) continue;                                          // :116-117
        const double slope = (double)(y2 - y1) / (double)(x2 - x1);
        if (fabs(slope) < 0.3) continue;                                 // :122-123
        const double mid = (double)(x1 + x2) / 2.0;
        const bool is_left = slope < 0 && mid < cx, is_right = slope > 0 && mid > cx;
        if ((side == 0 && !is_left) || (side == 1 && !is_right)) continue;
        ++nl;
        for (int e = 0; e < 2; ++e) {
            const double y = e ? y2 : y1, x = e ? x2 : x1;
            const double yy = y * y;
            s4 += yy * yy, s3 += yy * y, s2 += yy, s1 += y, s0 += 1.0;
            t2 += yy * x, t1 += y * x, t0 += x;
        }
    }
    double* st = lane_state + (size_t)s * 8 + side * 4;                  // c2 c1 c0 has_prev
    int* inf = info + (size_t)s * 8;
    if (side == 0) {
        inf[4] = n, inf[5] = npts[s], inf[6] = (int)thr[(size_t)s * 4], inf[7] = (int)thr[(size_t)s * 4 + 1];
    }
    inf[side] = 0, inf[2 + side] = nl;
    conf[(size_t)s * 2 + side] = 0.0;
    if (nl == 0) return;
    // scaled normal equations G = D^-1 A^T A D^-1, D = column norms (numpy: lhs /= scale)
    const double d0 = sqrt(s4), d1 = sqrt(s2), d2 = sqrt(s0);
    double G[3][3] = {{1.0, s3 / (d0 * d1), s2 / (d0 * d2)}, {s3 / (d0 * d1), 1.0, s1 / (d1 * d2)},
                      {s2 / (d0 * d2), s1 / (d1 * d2), 1.0}};
    if (d0 == 0.0) G[0][0] = 0.0, G[0][1] = G[1][0] = G[0][2] = G[2][0] = 0.0;
    if (d1 == 0.0) G[1][1] = 0.0, G[0][1] = G[1][0] = G[1][2] = G[2][1] = 0.0;
    const double rhs[3] = {d0 > 0 ? t2 / d0 : 0.0, d1 > 0 ? t1 / d1 : 0.0, t0 / d2};
    double V[3][3];
    jacobi3(G, V);
    double lmax = fmax(G[0][0], fmax(G[1][1], G[2][2]));
    const double rcond = (double)(2 * nl) * 2.220446049250313e-16;       // len(x) * eps
    double sol[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        const double lam = G[k][k];
        // lstsq drops singular values <= rcond * s_max.  Working on the Gram matrix squares the spectrum, so
        // an exactly dependent direction shows up as |lambda| ~ eps * lambda_max rather than 0: anything below
        // 1e-12 * lambda_max (singular value ratio 1e-6) is treated as dependent as well.
        if (!(lam > 1e-12 * lmax) || sqrt(lam) <= rcond * sqrt(lmax)) continue;
        const double proj = (V[0][k] * rhs[0] + V[1][k] * rhs[1] + V[2][k] * rhs[2]) / lam;
        for (int r = 0; r < 3; ++r) sol[r] += V[r][k] * proj;
    }
    double c2 = d0 > 0 ? sol[0] / d0 : 0.0, c1 = d1 > 0 ? sol[1] / d1 : 0.0, c0 = sol[2] / d2;
    if (st[3] != 0.0) {                                                  // :159-161
        c2 = smoothing * st[0] + (1.0 - smoothing) * c2;
        c1 = smoothing * st[1] + (1.0 - smoothing) * c1;
        c0 = smoothing * st[2] + (1.0 - smoothing) * c0;
    }
    st[0] = c2, st[1] = c1, st[2] = c0, st[3] = 1.0;
    double* po = poly + ((size_t)s * 2 + side) * 3;
    po[0] = c2, po[1] = c1, po[2] = c0;
    coef[0] = c2, coef[1] = c1, coef[2] = c0, coef[3] = 1.0;
    inf[side] = 1;
    conf[(size_t)s * 2 + side] = fmin(1.0, (double)nl / 10.0);          // :172
}

// one wave per (stream, side): lane 0 does the fit, then lane k makes point k of the 50
__global__ void __launch_bounds__(64) lane_fit_kernel(int S, int h, int w, int max_segments, double smoothing, const int* __restrict__ segs,
                                const int* __restrict__ nseg, double* __restrict__ lane_state,
                                double* __restrict__ poly, int* __restrict__ pts, int* __restrict__ info,
                                double* __restrict__ conf, const double* __restrict__ thr, const int* __restrict__ npts) {
    const int id = blockIdx.x, lane = threadIdx.x;
    if (id >= S * 2) return;
    const int s = id >> 1, side = id & 1;
    __shared__ double coef[4];                                           // c2 c1 c0 valid
    if (lane == 0) coef[3] = 0.0;
    if (lane == 0) lane_fit_one(s, side, h, w, max_segments, smoothing, segs, nseg, lane_state, poly, info, conf, thr, npts, coef);
    __syncthreads();
    if (coef[3] == 0.0 || lane >= 50) return;
    // np.linspace(h*0.6, h, 50); np.polyval (Horner); astype(int32) truncates toward zero
    const double c2 = coef[0], c1 = coef[1], c0 = coef[2];
    const double ya = (double)h * 0.6, yb = (double)h, step = (yb - ya) / 49.0;
    int* pp = pts + ((size_t)s * 2 + side) * 100;
    const double y = lane == 49 ? yb : (double)lane * step + ya;
    const double x = (c2 * y + c1) * y + c0;
    pp[2 * lane] = (int)x, pp[2 * lane + 1] = (int)y;
}

// the generic PPHT kernel alone (stage bit 3: tests of the fallback chain)
__global__ void __launch_bounds__(192) houghp_kernel(uint8_t* __restrict__ masked, int h, int w, int numrho, HoughCfg cfg,
                                                     unsigned* __restrict__ nz_all, const int* __restrict__ npts, int* __restrict__ accum_all,
                                                     const float* __restrict__ trig, int* __restrict__ segs, int* __restrict__ nseg,
                                                     const int* __restrict__ fallback, int* __restrict__ path, int rebuild_mask) {
    houghp_generic_body(masked, h, w, numrho, cfg, nz_all, npts, accum_all, trig, segs, nseg, fallback, path, rebuild_mask);
}

// The end of the lane chain as ONE launch, workgroup = frame: the single-workgroup PPHT for a frame the sharded kernel gave up, the
// generic PPHT for a frame that one cannot hold either (both leave at once for a frame that is done: the usual case), then the two
// fits (wave = side).  Three launches of ~4.7 + 4.7 + 9.5 us before.
__global__ void __launch_bounds__(192) hough_tail_kernel(uint8_t* __restrict__ masked, int h, int w, int numrho, HoughCfg cfg,
                                                         unsigned* __restrict__ nz_all, const int* __restrict__ npts, int* __restrict__ accum_all,
                                                         const float* __restrict__ trig, int* __restrict__ segs, int* __restrict__ nseg,
                                                         int* __restrict__ fallback, int* __restrict__ path, int check_flag, int rebuild_mask,
                                                         int max_segments, double smoothing, double* __restrict__ lane_state,
                                                         double* __restrict__ poly, int* __restrict__ pts, int* __restrict__ info,
                                                         double* __restrict__ conf, const double* __restrict__ thr) {
    houghp_fast_body(h, w, numrho, cfg, nz_all, npts, accum_all, trig, segs, nseg, fallback, check_flag, path);
    __threadfence();
    __syncthreads();
    houghp_generic_body(masked, h, w, numrho, cfg, nz_all, npts, accum_all, trig, segs, nseg, fallback, path, rebuild_mask);
    __threadfence();
    __syncthreads();
    const int s = blockIdx.x, side = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ double coef2[2][4];                                       // c2 c1 c0 valid, per side
    if (side < 2 && lane == 0) {
        coef2[side][3] = 0.0;
        lane_fit_one(s, side, h, w, max_segments, smoothing, segs, nseg, lane_state, poly, info, conf, thr, npts, coef2[side]);
    }
    __syncthreads();
    if (side >= 2 || coef2[side][3] == 0.0 || lane >= 50) return;
    const double c2 = coef2[side][0], c1 = coef2[side][1], c0 = coef2[side][2];
    const double ya = (double)h * 0.6, yb = (double)h, step = (yb - ya) / 49.0;
    int* pp = pts + ((size_t)s * 2 + side) * 100;
    const double y = lane == 49 ? yb : (double)lane * step + ya;
    const double x = (c2 * y + c1) * y + c0;
    pp[2 * lane] = (int)x, pp[2 * lane + 1] = (int)y;
}

struct LaneCtx {
    float* d_trig = nullptr;
    int* d_roi = nullptr;           // default trapezoid's [xl, xr] per row for frames of roi_h x roi_w (the bit-map resolve path)
    int roi_h = 0, roi_w = 0;
};

}  // namespace

extern "C" {

int av_lane_ctx_free(av_ctx* ctx) {
    if (ctx && ctx->lane) {
        LaneCtx* lc = (LaneCtx*)ctx->lane;
        if (lc->d_trig) (void)hipFree(lc->d_trig);
        if (lc->d_roi) (void)hipFree(lc->d_roi);
        delete lc;
        ctx->lane = nullptr;
    }
    return AV_OK;
}

size_t av_lane_workspace_bytes(int n_streams, int h, int w, int max_segments) {
    if (n_streams <= 0 || h <= 0 || w <= 0 || max_segments <= 0) return 0;
    return lane_layout(n_streams, h, w, max_segments).total;
}

int av_lane_workspace_view(int what, int n_streams, int h, int w, int max_segments, size_t* offset, size_t* bytes) {
    AV_REQUIRE(offset && bytes, AV_EINVAL, "av_lane_workspace_view: null out pointer");
    const LaneWs L = lane_layout(n_streams, h, w, max_segments);
    const size_t px = (size_t)n_streams * h * w;
    switch (what) {
        case 0: *offset = L.blur, *bytes = px; break;
        case 1: *offset = L.map, *bytes = px; break;
        case 2: *offset = L.edges, *bytes = px; break;
        case 3: *offset = L.masked, *bytes = px; break;
        case 4: *offset = L.thr, *bytes = (size_t)n_streams * 32; break;
        case 5: *offset = L.segs, *bytes = (size_t)n_streams * max_segments * 16; break;
        case 6: *offset = L.nseg, *bytes = (size_t)n_streams * 4; break;
        case 7: *offset = L.accum, *bytes = (size_t)n_streams * NUMANGLE * L.numrho * 4; break;
        case 8: *offset = L.rowcnt + (size_t)n_streams * 4, *bytes = (size_t)n_streams * 4; break;     // Hough kernel taken per frame
        case 9: *offset = L.nz, *bytes = px * 4; break;                      // point lists: frame s at [s * h * w], x | y << 16, row-major
        case 10: *offset = L.npts, *bytes = (size_t)n_streams * 4; break;
        default: av_set_error("av_lane_workspace_view: unknown view %d", what); return AV_EINVAL;
    }
    return AV_OK;
}

int av_lane_workspace_init(av_ctx* ctx, av_stream_t stream, int n_streams, int h, int w, int max_segments, void* ws) {
    AV_REQUIRE(ctx && ws, AV_EINVAL, "av_lane_workspace_init: null argument");
    const LaneWs L = lane_layout(n_streams, h, w, max_segments);
    AV_HIP(hipMemsetAsync(ws, 0, L.total, as_stream(stream)));     // histogram, accumulator start at zero
    return AV_OK;
}

int av_lane_detect(av_ctx* ctx, av_stream_t stream, const av_lane_cfg* cfg, int n_streams, int h, int w,
                   const uint8_t* bgr, const int32_t* roi_rows, void* workspace, double* lane_state, double* poly,
                   int32_t* pts, int32_t* info, double* conf, int stages) {
    AV_REQUIRE(ctx && cfg && bgr && workspace && lane_state && poly && pts && info && conf, AV_EINVAL,
               "av_lane_detect: null argument");
    AV_REQUIRE(n_streams > 0 && h >= 8 && w >= 8 && h < 32768 && w < 32768, AV_EINVAL, "av_lane_detect: bad frame size %dx%d", w, h);
    AV_REQUIRE(cfg->max_segments > 0 && cfg->hough_threshold > 0, AV_EINVAL, "av_lane_detect: bad configuration");
    AV_REQUIRE((size_t)h * w < (1u << 31), AV_EINVAL, "av_lane_detect: frame too large for 31-bit labels");
    hipStream_t st = as_stream(stream);
    if (!ctx->lane) {
        // trig table of cv::HoughLinesProbabilistic: theta is a float, angles n*theta in double
        LaneCtx* lc = new (std::nothrow) LaneCtx();
        AV_REQUIRE(lc, AV_ENOMEM, "av_lane_detect: out of host memory");
        float tt[2 * NUMANGLE];
        const float theta = (float)(3.14159265358979323846 / 180.0);
        for (int n = 0; n < NUMANGLE; ++n) {
            tt[2 * n] = (float)(std::cos((double)n * theta) * 1.0f);
            tt[2 * n + 1] = (float)(std::sin((double)n * theta) * 1.0f);
        }
        AV_HIP(hipMalloc(&lc->d_trig, sizeof(tt)));
        AV_HIP(hipMemcpy(lc->d_trig, tt, sizeof(tt), hipMemcpyHostToDevice));
        ctx->lane = lc;
    }
    LaneCtx* lc = (LaneCtx*)ctx->lane;
    const LaneWs L = lane_layout(n_streams, h, w, cfg->max_segments);
    unsigned char* ws = (unsigned char*)workspace;
    uint8_t* blur = ws + L.blur;
    uint8_t* map = ws + L.map;
    unsigned* labels = (unsigned*)(ws + L.labels);
    uint8_t* edges = ws + L.edges;
    uint8_t* masked = ws + L.masked;
    unsigned* hist = (unsigned*)(ws + L.hist);
    double* thr = (double*)(ws + L.thr);
    int* rowcnt = (int*)(ws + L.rowcnt);
    unsigned* nz = (unsigned*)(ws + L.nz);
    int* npts = (int*)(ws + L.npts);
    int* accum = (int*)(ws + L.accum);
    int* segs = (int*)(ws + L.segs);
    int* nseg = (int*)(ws + L.nseg);
    unsigned* tedge = (unsigned*)(ws + L.tedge);
    unsigned* rbits = (unsigned*)(ws + L.rbits);
    unsigned* kbits = (unsigned*)(ws + L.kbits);
    bool bitpath = false;            // this call resolved the ROI through the bit maps (no masked byte map written)
    bool prepped = false;            // the compaction pass cleared the sharded Hough kernel's exchange words and fallback flags
    const char* shard_env = getenv("AVHOT_HOUGH_SHARD");
    const bool use_shard = !(stages & 8) && !(shard_env && atoi(shard_env) == 0);      // AVHOT_HOUGH_SHARD=0 skips the sharded kernel
    const dim3 tiles((w + TW - 1) / TW, (h + TH - 1) / TH, n_streams);
    const bool fastp = (w % 16 == 0) && w >= 32 && (((size_t)bgr | (size_t)workspace) & 15) == 0 &&
                       (long long)h * (w >> 4) < (1ll << 24);             // chunk_xy's exact range
    const bool streamp = (w % 4 == 0) && w >= 8 && (((size_t)bgr | (size_t)workspace) & 15) == 0 && !(stages & 4);
    if (!(stages & 16)) {                                          // bit 4: Hough + fit only, on the point lists already in the workspace
        // fused = one streaming pass BGR -> non-maximum-suppressed magnitudes (thresholds applied by the hysteresis pass);
        // other shapes take the two-pass kernels with the blurred image in memory between them
        const bool fused = streamp && fastp && !getenv("AVHOT_LANE_TWO_PASS");
        if (fused) {
            const char* fe = getenv("AVHOT_LANE_FROWS");
            const int fr = fe ? atoi(fe) : 48;                       // front_stream at 720p, 64 frames: 45 rows 122 us, 72 rows 128, 90 rows 123
            // a few frames per launch (the per-frame class calls): short bands, so that a frame is hundreds of waves instead of 96
            const int frows = (stages & 1) ? 72 : (fe ? (fr == 72 ? 72 : (fr == 90 ? 90 : (fr == 15 ? 15 : (fr == 48 ? 48 : 45)))) : (n_streams <= 8 ? 15 : 48));
            const int sfrows = frows == 48 ? 45 : frows;           // front_stream has no 48-row instantiation
            const dim3 fgrid((((w + SW - 1) / SW) * ((h + sfrows - 1) / sfrows) + 3) / 4, 1, n_streams);     // front_stream: waves = strips x bands
            const bool packed = (unsigned long long)n_streams * h * w * 3ull < (1ull << 32) && !getenv("AVHOT_LANE_STRIP_FRONT");
            if (packed) {
                // work geometry of front_pack: full strips of 62 chunks, the remainder chunks of G frames share one wave
                FrontGeo g{};
                const int C = w >> 2;
                g.S = n_streams, g.nb = (h + frows - 1) / frows, g.nfull = C / 62, g.rem = C - 62 * g.nfull;
                g.G = g.rem ? 64 / (g.rem + 2) : 0;
                if (g.G > 16) g.G = 16;
                g.nc = g.G > 6 ? g.G : 6;                              // 6 KB of LDS per wave: 26 waves fit a CU
                const int ngr = g.G ? (n_streams + g.G - 1) / g.G : 0;
                const dim3 pgrid((unsigned)(n_streams * g.nb * g.nfull + ngr * g.nb));      // one wave per workgroup
                const size_t lds = (size_t)g.nc * 1024;
                const bool timed = getenv("AVHOT_LANE_TIMED") != nullptr && !(stages & 1);
                if (stages & 1) hipLaunchKernelGGL((front_pack<true, 72, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (timed && frows == 48 && getenv("AVHOT_ABL") && atoi(getenv("AVHOT_ABL")) == 1) hipLaunchKernelGGL((front_pack<false, 48, true, 1>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (timed && frows == 48 && getenv("AVHOT_ABL") && atoi(getenv("AVHOT_ABL")) == 2) hipLaunchKernelGGL((front_pack<false, 48, true, 2>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (timed && frows == 48 && getenv("AVHOT_ABL") && atoi(getenv("AVHOT_ABL")) == 6) hipLaunchKernelGGL((front_pack<false, 48, true, 6>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (timed && frows == 48 && getenv("AVHOT_ABL") && atoi(getenv("AVHOT_ABL")) == 7) hipLaunchKernelGGL((front_pack<false, 48, true, 7>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (timed && frows == 48) hipLaunchKernelGGL((front_pack<false, 48, true>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (frows == 48) hipLaunchKernelGGL((front_pack<false, 48, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (frows == 45) hipLaunchKernelGGL((front_pack<false, 45, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (frows == 90) hipLaunchKernelGGL((front_pack<false, 90, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else if (frows == 15) hipLaunchKernelGGL((front_pack<false, 15, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
                else hipLaunchKernelGGL((front_pack<false, 72, false>), pgrid, dim3(64), lds, st, bgr, h, w, g, blur, map, hist);
            } else
            if (stages & 1) hipLaunchKernelGGL((front_stream<true, 72>), fgrid, dim3(256), 0, st, bgr, h, w, blur, map, hist);
            else if (frows == 45 || frows == 48) hipLaunchKernelGGL((front_stream<false, 45>), fgrid, dim3(256), 0, st, bgr, h, w, blur, map, hist);
            else if (frows == 90) hipLaunchKernelGGL((front_stream<false, 90>), fgrid, dim3(256), 0, st, bgr, h, w, blur, map, hist);
            else if (frows == 15) hipLaunchKernelGGL((front_stream<false, 15>), fgrid, dim3(256), 0, st, bgr, h, w, blur, map, hist);
            else hipLaunchKernelGGL((front_stream<false, 72>), fgrid, dim3(256), 0, st, bgr, h, w, blur, map, hist);
            AV_LAUNCH_CHECK();
            hipLaunchKernelGGL(thresholds_kernel, dim3(n_streams), dim3(256), 0, st, h, w, hist, thr, rowcnt, npts, nseg);
            AV_LAUNCH_CHECK();
        } else {
            const dim3 sgrid((((w + SW - 1) / SW) * ((h + SROWS - 1) / SROWS) + 3) / 4, 1, n_streams);      // waves = strips x bands
            if (streamp) hipLaunchKernelGGL(gray_blur_hist_stream, sgrid, dim3(256), 0, st, bgr, h, w, blur, hist);
            else if (fastp) hipLaunchKernelGGL(gray_blur_hist_fast, tiles, dim3(256), 0, st, bgr, h, w, blur, hist);
            else hipLaunchKernelGGL(gray_blur_hist_kernel, tiles, dim3(256), 0, st, bgr, h, w, blur, hist);
            AV_LAUNCH_CHECK();
            hipLaunchKernelGGL(thresholds_kernel, dim3(n_streams), dim3(256), 0, st, h, w, hist, thr, rowcnt, npts, nseg);
            AV_LAUNCH_CHECK();
            if (streamp) hipLaunchKernelGGL(sobel_nms_stream, sgrid, dim3(256), 0, st, blur, h, w, thr, map, labels);
            else if (fastp) hipLaunchKernelGGL(sobel_nms_fast, tiles, dim3(256), 0, st, blur, h, w, thr, map, labels);
            else hipLaunchKernelGGL(sobel_nms_kernel, tiles, dim3(256), 0, st, blur, h, w, thr, map, labels);
            AV_LAUNCH_CHECK();
        }
        Roi roi;
        roi.x0 = (int)(w * 0.1), roi.x1 = (int)(w * 0.4), roi.x2 = (int)(w * 0.6), roi.x3 = (int)(w * 0.9);
        roi.yt = (int)(h * 0.6);                                       // lane_detector.py:55-60
        int cbox[4] = {0, 0, w >> 4, h}, cbox_rows = h;               // chunk box of the resolve / compaction passes
        BitBox bb{0, 0, 0, 0, 0};
        if (fastp) {
            // default ROI, no debug edge map: the tile pass leaves the ROI's candidates as one bit per pixel of the ROI's chunk box and
            // the resolve / compaction passes work on bit maps (AVHOT_LANE_BYTE_RESOLVE=1: the byte-map passes of round 3)
            bitpath = !(stages & 1) && !roi_rows && roi.yt < h && (h - roi.yt) <= CB_ROWS && !getenv("AVHOT_LANE_BYTE_RESOLVE");
            if (bitpath) {
                if (!lc->d_roi || lc->roi_h != h || lc->roi_w != w) {
                    std::vector<int> tab(2 * (size_t)h);
                    for (int y = 0; y < h; ++y) {
                        int xl = 1, xr = 0;
                        if (y >= roi.yt) {                             // roi_bounds' arithmetic
                            const long long den = h - roi.yt, t = h - y;
                            xl = (int)((2 * (roi.x0 * den + (long long)(roi.x1 - roi.x0) * t) + den) / (2 * den));
                            xr = (int)((2 * (roi.x3 * den + (long long)(roi.x2 - roi.x3) * t) + den) / (2 * den));
                        }
                        tab[2 * y] = xl, tab[2 * y + 1] = xr;
                    }
                    if (lc->d_roi) AV_HIP(hipFree(lc->d_roi));
                    lc->d_roi = nullptr;
                    AV_HIP(hipMalloc(&lc->d_roi, tab.size() * sizeof(int)));
                    AV_HIP(hipMemcpy(lc->d_roi, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
                    lc->roi_h = h, lc->roi_w = w;
                }
                bb.by0 = roi.yt, bb.bch = h - roi.yt;
                bb.bx0 = roi.x0 >> 4, bb.bcw = (((roi.x3 < w ? roi.x3 : w - 1) >> 4) - bb.bx0) + 1;
                bb.pw = (bb.bcw + 1) / 2;
            }
            const dim3 tgrid((w + CT_C - 1) / CT_C, ((h + CT_R - 1) / CT_R + CT_STACK - 1) / CT_STACK, n_streams);
            const int* rt = bitpath ? lc->d_roi : nullptr;
            if (fused) hipLaunchKernelGGL(ccl_tile_kernel<true>, tgrid, dim3(256), 0, st, map, h, w, thr, labels, tedge, rt, (uint16_t*)rbits, bb);
            else hipLaunchKernelGGL(ccl_tile_kernel<false>, tgrid, dim3(256), 0, st, map, h, w, thr, labels, tedge, rt, (uint16_t*)rbits, bb);
            const int nbh = (h - 1) / CT_R, nbv = (w - 1) / CT_C, span = (w > h ? w : h);
            if (nbh + nbv > 0) {
                const dim3 bgrid((span + 255) / 256, nbh + nbv, n_streams);
                hipLaunchKernelGGL(ccl_border_kernel, bgrid, dim3(256), 0, st, h, w, nbh, tedge, (int)tgrid.x, (int)tgrid.y, labels);
            }
            AV_LAUNCH_CHECK();
            // chunk box the resolve pass visits: everything for the debug edge map or a caller-defined ROI, else the
            // default trapezoid's bounding box (rows yt .. h-1, columns x0 .. x3)
            int bx0 = 0, by0 = 0, bcw = w >> 4, bch = h;
            if (!(stages & 1) && !roi_rows) {
                by0 = roi.yt < h ? roi.yt : h - 1, bch = h - by0;
                bx0 = roi.x0 >> 4, bcw = (((roi.x3 < w ? roi.x3 : w - 1) >> 4) - bx0) + 1;
            }
            cbox[0] = bx0, cbox[1] = by0, cbox[2] = bcw, cbox[3] = bch, cbox_rows = bch;
            const unsigned fchunks = (unsigned)bch * (unsigned)bcw;
            const dim3 ngrid((fchunks + 256 * FCK - 1) / (256 * FCK), n_streams);
            if (bitpath)
                hipLaunchKernelGGL(resolve_bits_kernel, dim3(((unsigned)(bb.bch * bb.pw) + 255) / 256, n_streams), dim3(256), 0, st, h, w,
                                   labels, rbits, kbits, rowcnt, bb);
            else if (fused)
                hipLaunchKernelGGL(finalize_fast<true>, ngrid, dim3(256), 0, st, map, h, w, thr, labels, roi, roi_rows,
                                   (stages & 1) ? edges : nullptr, masked, rowcnt, bx0, by0, bcw, bch);
            else
                hipLaunchKernelGGL(finalize_fast<false>, ngrid, dim3(256), 0, st, map, h, w, thr, labels, roi, roi_rows,
                                   (stages & 1) ? edges : nullptr, masked, rowcnt, bx0, by0, bcw, bch);
        } else {
            hipLaunchKernelGGL(ccl_merge_kernel, dim3((w + 63) / 64, (h + 3) / 4, n_streams), dim3(256), 0, st, map, h, w, labels);
            AV_LAUNCH_CHECK();
            hipLaunchKernelGGL(finalize_kernel, dim3((w + 1023) / 1024, h, n_streams), dim3(256), 0, st, map, h, w, labels, roi,
                               roi_rows, (stages & 1) ? edges : nullptr, masked, rowcnt);
        }
        AV_LAUNCH_CHECK();
        // (the fallback flags overlay the first n_streams row counters of frame 0: cleared by this pass only when those lie above the
        // box rows it reads, and only when the Hough stage follows in this call)
        const bool prep_here = bitpath && use_shard && !(stages & 2) && n_streams <= cbox[1] && fastp && cbox_rows <= CB_ROWS;
        if (fastp && cbox_rows <= CB_ROWS)
            hipLaunchKernelGGL(compact_box_kernel, dim3(n_streams, 4), dim3(1024), (size_t)cbox_rows * sizeof(int), st, masked, h, w,
                               rowcnt, nz, npts, cbox[0], cbox[1], cbox[2], cbox[3], bitpath ? kbits : nullptr, bb.pw,
                               prep_here ? accum : nullptr, (size_t)NUMANGLE * L.numrho, 8 * HG, rowcnt);
        else hipLaunchKernelGGL(compact_kernel, dim3(h, n_streams), dim3(256), 0, st, masked, h, w, rowcnt, nz, npts);
        AV_LAUNCH_CHECK();
        prepped = prep_here;
    }
    if (stages & 2) return AV_OK;                                  // pixel stages only (tests, profiling)
    HoughCfg hc{cfg->hough_threshold, cfg->min_line_length, cfg->max_line_gap, cfg->max_segments};
    if (stages & 32) {                                             // bit 5 (with bit 4): fit only, on the segments in the workspace
        hipLaunchKernelGGL(lane_fit_kernel, dim3(n_streams * 2), dim3(64), 0, st, n_streams, h, w, cfg->max_segments,
                           cfg->smoothing_factor, segs, nseg, lane_state, poly, pts, info, conf, thr, npts);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    // Hough-only calls (stage bit 4) follow a pixel-stage call of the same shape: the same path decision
    const bool rebuild = (stages & 16) ? (fastp && !roi_rows && (int)(h * 0.6) < h && (h - (int)(h * 0.6)) <= CB_ROWS && !getenv("AVHOT_LANE_BYTE_RESOLVE"))
                                       : bitpath;
    int* fb = rowcnt;       // the per-row counters are dead after compaction: their first n_streams words are the fallback flags,
    int* hpath = rowcnt + n_streams;   // the next n_streams say which kernel made a frame's segments (1 shard, 2 fast, 3 generic)
    const bool use_fast = !(stages & 8);
    if (use_fast) {
        // theta-sharded LDS variant first (4 workgroups per frame); frames it cannot hold, or where a partner did not
        // show up in time, are flagged for houghp_fast, and what that cannot hold for houghp_kernel.
        // AVHOT_HOUGH_SHARD=0 skips the first stage.
        if (use_shard) {
            if (!prepped) {
                hipLaunchKernelGGL(hough_prep_kernel, dim3((n_streams * 8 * HG + 255) / 256), dim3(256), 0, st, n_streams, L.numrho, accum, fb);
                AV_LAUNCH_CHECK();
            }
            const char* sp = getenv("AVHOT_HOUGH_SPIN");
            const char* dr = getenv("AVHOT_HOUGH_DROP");
            const int spin = sp && atoi(sp) > 0 ? atoi(sp) : HS_SPIN;
            hipLaunchKernelGGL(houghp_shard, dim3(n_streams * HG), dim3((HVW + 1) * 64), 0, st, h, w, L.numrho, hc, nz, npts, accum,
                               lc->d_trig, segs, nseg, fb, spin, dr ? atoi(dr) : -1, hpath, getenv("AVHOT_HOUGH_TIMED") ? 1 : 0,
                               (getenv("AVHOT_HOUGH_XCD") && atoi(getenv("AVHOT_HOUGH_XCD")) == 0) ? 0 : 1);
            AV_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(hough_tail_kernel, dim3(n_streams), dim3(192), 0, st, masked, h, w, L.numrho, hc, nz, npts, accum, lc->d_trig,
                           segs, nseg, fb, hpath, use_shard ? 1 : 0, rebuild ? 1 : 0, cfg->max_segments, cfg->smoothing_factor, lane_state,
                           poly, pts, info, conf, thr);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    hipLaunchKernelGGL(houghp_kernel, dim3(n_streams), dim3(192), 0, st, masked, h, w, L.numrho, hc, nz, npts, accum,
                       lc->d_trig, segs, nseg, use_fast ? fb : nullptr, hpath, rebuild ? 1 : 0);
    AV_LAUNCH_CHECK();
    hipLaunchKernelGGL(lane_fit_kernel, dim3(n_streams * 2), dim3(64), 0, st, n_streams, h, w,
                       cfg->max_segments, cfg->smoothing_factor, segs, nseg, lane_state, poly, pts, info, conf, thr, npts);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"
