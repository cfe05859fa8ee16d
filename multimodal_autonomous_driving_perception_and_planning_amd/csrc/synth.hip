// On-device synthetic road frames.
//
// Stands in for the reference's lost data/generators/synthetic_data.py (only a Python 3.12 .pyc of it
// survives): sky gradient, textured asphalt, two dashed lane lines converging to a vanishing
// point, a few box "vehicles".  Pure integer arithmetic, bit-identical to oracle/lane_ref.py:
// synthetic_frame(), so CPU oracle and GPU path see the same pixels without a PCIe copy.
#include "common.h"

namespace {

__device__ __forceinline__ int floordiv(int a, int b) {      // Python's // for b > 0
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

__global__ void __launch_bounds__(256) synth_kernel(int S, int h, int w, int stream0, int frame, uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)S * h * w) return;
    const int s = (int)(i / ((long long)h * w)), r = (int)(i - (long long)s * h * w), y = r / w, x = r - y * w;
    const int stream = stream0 + s;
    const int hz = (h * 9) / 20, hzd = hz > 1 ? hz : 1;
    int c0, c1, c2;
    if (y < hz) {
        c0 = 230 - (y * 60) / hzd, c1 = 190 - (y * 50) / hzd, c2 = 150 - (y * 70) / hzd;
    } else {
        const unsigned K = (unsigned)stream * 83492791u + (unsigned)frame * 2654435761u;
        unsigned hsh = ((unsigned)x * 73856093u) ^ ((unsigned)y * 19349663u) ^ K;
        hsh = (hsh ^ (hsh >> 13)) * 1274126177u;
        const int tex = (int)((hsh >> 24) & 15u);
        c0 = 84 + tex + 4, c1 = 84 + tex + 2, c2 = 84 + tex;
        const int den = (h - hz) > 1 ? (h - hz) : 1, t = y - hz;
        const int sway = ((stream * 7 + frame) % 32) - 16;
        const bool dash = (((y + 5 * frame) / 24) % 2) == 0;
        const int half = 1 + (6 * t) / den;
        const int xt[2] = {(w * 9) / 20, (w * 11) / 20}, xb[2] = {(w * 3) / 20, (w * 17) / 20};
        for (int q = 0; q < 2; ++q) {
            const int xc = xt[q] + floordiv((xb[q] - xt[q] + sway) * t, den);
            if (dash && abs(x - xc) <= half) c0 = c1 = c2 = 235;
        }
    }
    const int k = 3 + ((stream * 5 + frame / 8) % 4);
    for (int v = 0; v < k; ++v) {
        const int sd = (stream * 131 + v * 977 + (frame / 8) * 31) & 0xFFFF;
        const int bw = 50 + (sd % 90), bh = 36 + ((sd >> 3) % 60);
        const int mw = (w - bw) > 1 ? (w - bw) : 1, mh = (h - hz - bh - 10) > 1 ? (h - hz - bh - 10) : 1;
        const int bx = (sd * 37 + v * 211 + frame * (3 + v)) % mw;
        const int by = hz + 10 + ((sd >> 5) % mh);
        if (x >= bx && x < bx + bw && y >= by && y < by + bh)
            c0 = 40 + (sd % 160), c1 = 40 + ((sd >> 4) % 160), c2 = 40 + ((sd >> 8) % 160);
    }
    uint8_t* o = out + i * 3;
    o[0] = (uint8_t)c0, o[1] = (uint8_t)c1, o[2] = (uint8_t)c2;
}

// Row-wise variant for w % 16 == 0: every integer division above is constant along a row or over the frame,
// so a workgroup (a few rows of one frame) computes the vehicle boxes and its rows' constants once into LDS;
// a thread then produces 16 pixels = 48 bytes = three 16-byte stores.  Same formulas, same bytes.
constexpr int SY_ROWS = 8;
__global__ void __launch_bounds__(256) synth_rows_kernel(int h, int w, int stream0, int frame, int G, int RB,
                                                         uint8_t* __restrict__ out) {
    __shared__ int sbox[8][8];          // x0, x1, y0, y1, c0, c1, c2
    __shared__ int srow[SY_ROWS][8];    // sky?, c0, c1, c2 | xc0, xc1, half (-1: no dash on this row), y hash term
    const int s = blockIdx.y, stream = stream0 + s, t = threadIdx.x, row0 = blockIdx.x * RB;
    const int hz = (h * 9) / 20, hzd = hz > 1 ? hz : 1;
    const int k = 3 + ((stream * 5 + frame / 8) % 4);
    if (t < k) {
        const int v = t;
        const int sd = (stream * 131 + v * 977 + (frame / 8) * 31) & 0xFFFF;
        const int bw = 50 + (sd % 90), bh = 36 + ((sd >> 3) % 60);
        const int mw = (w - bw) > 1 ? (w - bw) : 1, mh = (h - hz - bh - 10) > 1 ? (h - hz - bh - 10) : 1;
        const int bx = (sd * 37 + v * 211 + frame * (3 + v)) % mw;
        const int by = hz + 10 + ((sd >> 5) % mh);
        sbox[v][0] = bx, sbox[v][1] = bx + bw, sbox[v][2] = by, sbox[v][3] = by + bh;
        sbox[v][4] = 40 + (sd % 160), sbox[v][5] = 40 + ((sd >> 4) % 160), sbox[v][6] = 40 + ((sd >> 8) % 160);
    }
    if (t >= 64 && t < 64 + RB) {
        const int r = t - 64, y = row0 + r;
        if (y < h) {
            if (y < hz) {
                srow[r][0] = 1;
                srow[r][1] = 230 - (y * 60) / hzd, srow[r][2] = 190 - (y * 50) / hzd, srow[r][3] = 150 - (y * 70) / hzd;
            } else {
                const int den = (h - hz) > 1 ? (h - hz) : 1, tt = y - hz;
                const int sway = ((stream * 7 + frame) % 32) - 16;
                const bool dash = (((y + 5 * frame) / 24) % 2) == 0;
                const int xt0 = (w * 9) / 20, xt1 = (w * 11) / 20, xb0 = (w * 3) / 20, xb1 = (w * 17) / 20;
                srow[r][0] = 0;
                srow[r][4] = xt0 + floordiv((xb0 - xt0 + sway) * tt, den);
                srow[r][5] = xt1 + floordiv((xb1 - xt1 + sway) * tt, den);
                srow[r][6] = dash ? 1 + (6 * tt) / den : -1;
                srow[r][7] = (int)((unsigned)y * 19349663u);
            }
        }
    }
    __syncthreads();
    const int r = t / G, g = t - r * G, y = row0 + r;
    if (r >= RB || y >= h) return;
    const bool sky = srow[r][0] != 0;
    const int sc0 = srow[r][1], sc1 = srow[r][2], sc2 = srow[r][3];
    const int xc0 = srow[r][4], xc1 = srow[r][5], half = srow[r][6];
    const unsigned K = ((unsigned)stream * 83492791u + (unsigned)frame * 2654435761u) ^ (unsigned)srow[r][7];
    // vehicle boxes crossing this row, as x ranges (empty when the row misses the box)
    int bx0[6], bx1[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        const bool on = v < k && y >= sbox[v][2] && y < sbox[v][3];
        bx0[v] = on ? sbox[v][0] : 0, bx1[v] = on ? sbox[v][1] : 0;
    }
    unsigned pk[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) pk[q] = 0;
    bool any_box = false;                       // most rows cross no vehicle box: the six range tests per pixel are skipped there
#pragma unroll
    for (int v = 0; v < 6; ++v) any_box = any_box || bx1[v] > bx0[v];
    if (sky && !any_box) {                      // a sky row is one colour: 16 pixels = the 3-byte pattern repeated, no per-pixel work
        const unsigned t = (unsigned)sc0 | ((unsigned)sc1 << 8) | ((unsigned)sc2 << 16);
        const unsigned w0 = t | (t << 24), w1 = (t >> 8) | (t << 16), w2 = (t >> 16) | (t << 8);
#pragma unroll
        for (int q = 0; q < 12; q += 3) pk[q] = w0, pk[q + 1] = w1, pk[q + 2] = w2;
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int x = g * 16 + j;
            int c0, c1, c2;
            if (sky) {
                c0 = sc0, c1 = sc1, c2 = sc2;
            } else {
                unsigned hsh = ((unsigned)x * 73856093u) ^ K;
                hsh = (hsh ^ (hsh >> 13)) * 1274126177u;
                const int tex = (int)((hsh >> 24) & 15u);
                c0 = 84 + tex + 4, c1 = 84 + tex + 2, c2 = 84 + tex;
                if (half >= 0 && (abs(x - xc0) <= half || abs(x - xc1) <= half)) c0 = c1 = c2 = 235;
            }
            if (any_box) {
#pragma unroll
                for (int v = 0; v < 6; ++v)
                    if (x >= bx0[v] && x < bx1[v]) c0 = sbox[v][4], c1 = sbox[v][5], c2 = sbox[v][6];
            }
            const int b = j * 3;
            pk[b >> 2] |= (unsigned)c0 << (8 * (b & 3));
            pk[(b + 1) >> 2] |= (unsigned)c1 << (8 * ((b + 1) & 3));
            pk[(b + 2) >> 2] |= (unsigned)c2 << (8 * ((b + 2) & 3));
        }
    }
    uint4* o = reinterpret_cast<uint4*>(out + (((size_t)s * h + y) * w + (size_t)g * 16) * 3);
    o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    o[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    o[2] = make_uint4(pk[8], pk[9], pk[10], pk[11]);
}

}  // namespace

extern "C" int av_synth_frames(av_ctx* ctx, av_stream_t stream, int n_streams, int h, int w, int stream0, int frame,
                               uint8_t* bgr) {
    AV_REQUIRE(ctx && bgr, AV_EINVAL, "av_synth_frames: null argument");
    AV_REQUIRE(n_streams > 0 && h >= 32 && w >= 32 && frame >= 0 && stream0 >= 0, AV_EINVAL, "av_synth_frames: bad arguments");
    if (w % 16 == 0 && w / 16 <= 256 && (reinterpret_cast<uintptr_t>(bgr) & 15) == 0) {
        const int G = w / 16;
        int RB = 256 / G;
        RB = RB > SY_ROWS ? SY_ROWS : RB;
        hipLaunchKernelGGL(synth_rows_kernel, dim3((h + RB - 1) / RB, n_streams), dim3(256), 0, as_stream(stream), h, w,
                           stream0, frame, G, RB, bgr);
        AV_LAUNCH_CHECK();
        return AV_OK;
    }
    const long long n = (long long)n_streams * h * w;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n_streams, h, w,
                       stream0, frame, bgr);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
