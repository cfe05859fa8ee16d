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

}  // namespace

extern "C" int av_synth_frames(av_ctx* ctx, av_stream_t stream, int n_streams, int h, int w, int stream0, int frame,
                               uint8_t* bgr) {
    AV_REQUIRE(ctx && bgr, AV_EINVAL, "av_synth_frames: null argument");
    AV_REQUIRE(n_streams > 0 && h >= 32 && w >= 32 && frame >= 0 && stream0 >= 0, AV_EINVAL, "av_synth_frames: bad arguments");
    const long long n = (long long)n_streams * h * w;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n_streams, h, w,
                       stream0, frame, bgr);
    AV_LAUNCH_CHECK();
    return AV_OK;
}
