// f-2: BEV panel and camera-overlay rendering on the device.
//
// Reference: src/visualization/bev_renderer.py:286-348 (render; draw_vehicle :120-183, draw_agents :185-235,
// draw_trajectory :237-273, draw_uncertainty_ellipse :275-284, create_base_image :92-118, _draw_legend :350-364),
// src/visualization/overlays.py:26-210, and the draw_* methods of the detector / lane detector / tracker / planner
// (detector.py:171, lane_detector.py:220, multi_object_tracker.py:251, motion_planner.py:305).  All of that is
// cv2 drawing calls; OpenCV is not available here, so what is restated is the GEOMETRY each call draws (filled
// rectangles and convex quadrilaterals, thick segments, discs, circles, labels, 0.7/0.3 blends), rasterised by exact
// integer point-in-primitive tests per pixel -- not OpenCV's Bresenham / fixed-point scan conversion, and labels use a
// 5x7 bitmap font instead of Hershey Simplex.  PARITY UNPINNED against cv2; bit-exact against oracle/raster_ref.py.
//
// A picture is an ordered list of av_prim records (painter's algorithm: later primitives overwrite earlier ones).
// Workgroup = one 32x32 tile of one image: it filters the list down to the primitives whose bounding box touches the
// tile (order kept, LDS), then every thread walks that short list for its 4 pixels, which live in registers from the
// first read to the only write.  HBM traffic = the image once in, once out.
#include "common.h"

namespace {

constexpr int RT = 32;                     // tile edge
constexpr int RLIST = 1024;                // tile-local primitive indices held in LDS at a time

// 5x7 glyphs, ASCII 32..126, five column bytes each (bit 0 = top row): the classic public-domain LCD font
__constant__ unsigned char FONT5X7[95][5] = {
    {0x00, 0x00, 0x00, 0x00, 0x00}, {0x00, 0x00, 0x5F, 0x00, 0x00}, {0x00, 0x07, 0x00, 0x07, 0x00}, {0x14, 0x7F, 0x14, 0x7F, 0x14},
    {0x24, 0x2A, 0x7F, 0x2A, 0x12}, {0x23, 0x13, 0x08, 0x64, 0x62}, {0x36, 0x49, 0x55, 0x22, 0x50}, {0x00, 0x05, 0x03, 0x00, 0x00},
    {0x00, 0x1C, 0x22, 0x41, 0x00}, {0x00, 0x41, 0x22, 0x1C, 0x00}, {0x14, 0x08, 0x3E, 0x08, 0x14}, {0x08, 0x08, 0x3E, 0x08, 0x08},
    {0x00, 0x50, 0x30, 0x00, 0x00}, {0x08, 0x08, 0x08, 0x08, 0x08}, {0x00, 0x60, 0x60, 0x00, 0x00}, {0x20, 0x10, 0x08, 0x04, 0x02},
    {0x3E, 0x51, 0x49, 0x45, 0x3E}, {0x00, 0x42, 0x7F, 0x40, 0x00}, {0x42, 0x61, 0x51, 0x49, 0x46}, {0x21, 0x41, 0x45, 0x4B, 0x31},
    {0x18, 0x14, 0x12, 0x7F, 0x10}, {0x27, 0x45, 0x45, 0x45, 0x39}, {0x3C, 0x4A, 0x49, 0x49, 0x30}, {0x01, 0x71, 0x09, 0x05, 0x03},
    {0x36, 0x49, 0x49, 0x49, 0x36}, {0x06, 0x49, 0x49, 0x29, 0x1E}, {0x00, 0x36, 0x36, 0x00, 0x00}, {0x00, 0x56, 0x36, 0x00, 0x00},
    {0x08, 0x14, 0x22, 0x41, 0x00}, {0x14, 0x14, 0x14, 0x14, 0x14}, {0x00, 0x41, 0x22, 0x14, 0x08}, {0x02, 0x01, 0x51, 0x09, 0x06},
    {0x32, 0x49, 0x79, 0x41, 0x3E}, {0x7E, 0x11, 0x11, 0x11, 0x7E}, {0x7F, 0x49, 0x49, 0x49, 0x36}, {0x3E, 0x41, 0x41, 0x41, 0x22},
    {0x7F, 0x41, 0x41, 0x22, 0x1C}, {0x7F, 0x49, 0x49, 0x49, 0x41}, {0x7F, 0x09, 0x09, 0x09, 0x01}, {0x3E, 0x41, 0x49, 0x49, 0x7A},
    {0x7F, 0x08, 0x08, 0x08, 0x7F}, {0x00, 0x41, 0x7F, 0x41, 0x00}, {0x20, 0x40, 0x41, 0x3F, 0x01}, {0x7F, 0x08, 0x14, 0x22, 0x41},
    {0x7F, 0x40, 0x40, 0x40, 0x40}, {0x7F, 0x02, 0x0C, 0x02, 0x7F}, {0x7F, 0x04, 0x08, 0x10, 0x7F}, {0x3E, 0x41, 0x41, 0x41, 0x3E},
    {0x7F, 0x09, 0x09, 0x09, 0x06}, {0x3E, 0x41, 0x51, 0x21, 0x5E}, {0x7F, 0x09, 0x19, 0x29, 0x46}, {0x46, 0x49, 0x49, 0x49, 0x31},
    {0x01, 0x01, 0x7F, 0x01, 0x01}, {0x3F, 0x40, 0x40, 0x40, 0x3F}, {0x1F, 0x20, 0x40, 0x20, 0x1F}, {0x3F, 0x40, 0x38, 0x40, 0x3F},
    {0x63, 0x14, 0x08, 0x14, 0x63}, {0x07, 0x08, 0x70, 0x08, 0x07}, {0x61, 0x51, 0x49, 0x45, 0x43}, {0x00, 0x7F, 0x41, 0x41, 0x00},
    {0x02, 0x04, 0x08, 0x10, 0x20}, {0x00, 0x41, 0x41, 0x7F, 0x00}, {0x04, 0x02, 0x01, 0x02, 0x04}, {0x40, 0x40, 0x40, 0x40, 0x40},
    {0x00, 0x01, 0x02, 0x04, 0x00}, {0x20, 0x54, 0x54, 0x54, 0x78}, {0x7F, 0x48, 0x44, 0x44, 0x38}, {0x38, 0x44, 0x44, 0x44, 0x20},
    {0x38, 0x44, 0x44, 0x48, 0x7F}, {0x38, 0x54, 0x54, 0x54, 0x18}, {0x08, 0x7E, 0x09, 0x01, 0x02}, {0x0C, 0x52, 0x52, 0x52, 0x3E},
    {0x7F, 0x08, 0x04, 0x04, 0x78}, {0x00, 0x44, 0x7D, 0x40, 0x00}, {0x20, 0x40, 0x44, 0x3D, 0x00}, {0x7F, 0x10, 0x28, 0x44, 0x00},
    {0x00, 0x41, 0x7F, 0x40, 0x00}, {0x7C, 0x04, 0x18, 0x04, 0x78}, {0x7C, 0x08, 0x04, 0x04, 0x78}, {0x38, 0x44, 0x44, 0x44, 0x38},
    {0x7C, 0x14, 0x14, 0x14, 0x08}, {0x08, 0x14, 0x14, 0x18, 0x7C}, {0x7C, 0x08, 0x04, 0x04, 0x08}, {0x48, 0x54, 0x54, 0x54, 0x20},
    {0x04, 0x3F, 0x44, 0x40, 0x20}, {0x3C, 0x40, 0x40, 0x20, 0x7C}, {0x1C, 0x20, 0x40, 0x20, 0x1C}, {0x3C, 0x40, 0x30, 0x40, 0x3C},
    {0x44, 0x28, 0x10, 0x28, 0x44}, {0x0C, 0x50, 0x50, 0x50, 0x3C}, {0x44, 0x64, 0x54, 0x4C, 0x44}, {0x00, 0x08, 0x36, 0x41, 0x00},
    {0x00, 0x00, 0x7F, 0x00, 0x00}, {0x00, 0x41, 0x36, 0x08, 0x00}, {0x10, 0x08, 0x08, 0x10, 0x08}};

__device__ __forceinline__ void prim_bbox(const av_prim& q, int& bx0, int& by0, int& bx1, int& by1) {
    switch (q.type) {
        case AV_PRIM_SEG: {
            const int r = (q.p + 1) / 2;
            bx0 = min(q.x0, q.x1) - r, bx1 = max(q.x0, q.x1) + r, by0 = min(q.y0, q.y1) - r, by1 = max(q.y0, q.y1) + r;
            break;
        }
        case AV_PRIM_QUAD:
            bx0 = min(min(q.x0, q.x1), min(q.x2, q.x3)), bx1 = max(max(q.x0, q.x1), max(q.x2, q.x3));
            by0 = min(min(q.y0, q.y1), min(q.y2, q.y3)), by1 = max(max(q.y0, q.y1), max(q.y2, q.y3));
            break;
        case AV_PRIM_DISC:
        case AV_PRIM_RING:
            bx0 = q.x0 - q.p - 1, bx1 = q.x0 + q.p + 1, by0 = q.y0 - q.p - 1, by1 = q.y0 + q.p + 1;
            break;
        case AV_PRIM_GLYPH: {
            const int sc = q.x1 > 0 ? q.x1 : 1;
            bx0 = q.x0, by0 = q.y0, bx1 = q.x0 + 5 * sc - 1, by1 = q.y0 + 7 * sc - 1;
            break;
        }
        case AV_PRIM_POLY_BLEND:
            bx0 = q.x2, by0 = q.y2, bx1 = q.x3, by1 = q.y3;                // the builder stores the polygon's bounding box
            break;
        case AV_PRIM_RECT:
        case AV_PRIM_BLEND_RECT:
            bx0 = min(q.x0, q.x1), bx1 = max(q.x0, q.x1), by0 = min(q.y0, q.y1), by1 = max(q.y0, q.y1);
            break;
        default:                                                            // 0 = empty slot of a fixed-layout list
            bx0 = by0 = 0, bx1 = by1 = -1;
            break;
    }
}

// is pixel (x, y) painted by q?  Exact integer tests (the oracle does the same arithmetic in int64).
__device__ __forceinline__ bool prim_covers(const av_prim& q, int x, int y, const int32_t* __restrict__ verts) {
    switch (q.type) {
        case AV_PRIM_RECT:
        case AV_PRIM_BLEND_RECT:
            return x >= min(q.x0, q.x1) && x <= max(q.x0, q.x1) && y >= min(q.y0, q.y1) && y <= max(q.y0, q.y1);
        case AV_PRIM_SEG: {
            // distance from the pixel centre to the segment <= thickness / 2.  With t = (P - A).D: before A (t < 0) or
            // past B (t > |D|^2) it is the distance to that end point, else |cross(P - A, D)| / |D|; everything times 4 |D|^2
            // stays below 2^63 for image-sized coordinates (< 2^13)
            const long long dx = q.x1 - q.x0, dy = q.y1 - q.y0, px = x - q.x0, py = y - q.y0, th = q.p;
            const long long L2 = dx * dx + dy * dy, t = px * dx + py * dy;
            if (t <= 0) return 4 * (px * px + py * py) <= th * th;
            if (t >= L2) return 4 * ((px - dx) * (px - dx) + (py - dy) * (py - dy)) <= th * th;
            const long long cr = px * dy - py * dx;
            return 4 * cr * cr <= th * th * L2;
        }
        case AV_PRIM_QUAD: {
            const int xs[4] = {q.x0, q.x1, q.x2, q.x3}, ys[4] = {q.y0, q.y1, q.y2, q.y3};
            bool pos = true, neg = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int k1 = (k + 1) & 3;
                const long long c = (long long)(xs[k1] - xs[k]) * (y - ys[k]) - (long long)(ys[k1] - ys[k]) * (x - xs[k]);
                pos = pos && c >= 0, neg = neg && c <= 0;
            }
            return pos || neg;
        }
        case AV_PRIM_DISC: {
            const long long dx = x - q.x0, dy = y - q.y0, r = q.p;
            return dx * dx + dy * dy <= r * r;
        }
        case AV_PRIM_RING: {
            const long long dx = x - q.x0, dy = y - q.y0, r = q.p, d4 = 4 * (dx * dx + dy * dy);
            return d4 >= (2 * r - 1) * (2 * r - 1) && d4 <= (2 * r + 1) * (2 * r + 1);
        }
        case AV_PRIM_GLYPH: {
            const int sc = q.x1 > 0 ? q.x1 : 1;
            const int gx = x - q.x0, gy = y - q.y0;
            if (gx < 0 || gy < 0 || gx >= 5 * sc || gy >= 7 * sc || q.p < 32 || q.p > 126) return false;
            return (FONT5X7[q.p - 32][gx / sc] >> (gy / sc)) & 1;
        }
        case AV_PRIM_POLY_BLEND: {
            // even-odd rule on the vertex list verts[2 * (x0 + k)], k < y0; the pixel centre against half-open edges
            bool in = false;
            const int v0 = q.x0, n = q.y0;
            for (int k = 0, j = n - 1; k < n; j = k++) {
                const long long xi = verts[2 * (v0 + k)], yi = verts[2 * (v0 + k) + 1], xj = verts[2 * (v0 + j)], yj = verts[2 * (v0 + j) + 1];
                if ((yi > y) != (yj > y)) {
                    // x < xi + (xj - xi) (y - yi) / (yj - yi), multiplied out with the sign of (yj - yi)
                    const long long lhs = (x - xi) * (yj - yi), rhs = (xj - xi) * (y - yi);
                    if ((yj > yi) ? (lhs < rhs) : (lhs > rhs)) in = !in;
                }
            }
            return in;
        }
        default:
            return false;
    }
}

__device__ __forceinline__ unsigned prim_apply(const av_prim& q, unsigned px) {
    const unsigned c = (unsigned)q.b | ((unsigned)q.g << 8) | ((unsigned)q.r << 16);
    if (q.type == AV_PRIM_BLEND_RECT || q.type == AV_PRIM_POLY_BLEND) {
        // cv2.addWeighted(frame, 0.7, overlay, 0.3, 0) where the overlay is the frame with the shape painted: (7 p + 3 c + 5) / 10
        unsigned o = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned p = (px >> (8 * k)) & 255u, cc = (c >> (8 * k)) & 255u;
            o |= ((7u * p + 3u * cc + 5u) / 10u) << (8 * k);
        }
        return o;
    }
    return c;
}

__global__ void __launch_bounds__(256) raster_kernel(int n_images, int H, int W, uint8_t* __restrict__ img, const av_prim* __restrict__ prims,
                                                     int prim_cap, const int32_t* __restrict__ n_prims, const int32_t* __restrict__ verts,
                                                     int vert_cap) {
    __shared__ unsigned short list[RLIST];
    __shared__ int s_cnt, s_wbase[4];
    const int im = blockIdx.z, tx0 = blockIdx.x * RT, ty0 = blockIdx.y * RT, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const av_prim* pr = prims + (size_t)im * prim_cap;
    const int32_t* vt = verts ? verts + (size_t)im * vert_cap * 2 : nullptr;
    const int np = min(n_prims[im], prim_cap);
    uint8_t* base = img + (size_t)im * H * W * 3;
    // this thread's 4 pixels: row ty0 + tid / 8, columns tx0 + 4 (tid % 8) ..
    const int y = ty0 + (tid >> 3), xb = tx0 + 4 * (tid & 7);
    unsigned px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = 0;
        if (y < H && xb + k < W) {
            const uint8_t* p = base + ((size_t)y * W + xb + k) * 3;
            px[k] = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
        }
    }
    for (int c0 = 0; c0 < np;) {
        // ---- fill the tile list from primitives c0.. (order kept), until the list is full or the primitives run out ----
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int c = c0;
        for (; c < np; c += 256) {
            const int i = c + tid;
            bool hit = false;
            if (i < np) {
                int bx0, by0, bx1, by1;
                prim_bbox(pr[i], bx0, by0, bx1, by1);
                hit = bx1 >= tx0 && bx0 < tx0 + RT && by1 >= ty0 && by0 < ty0 + RT;
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) s_wbase[wid] = __popcll(bal);
            __syncthreads();
            int off = s_cnt;
            for (int q = 0; q < wid; ++q) off += s_wbase[q];
            const int tot = s_wbase[0] + s_wbase[1] + s_wbase[2] + s_wbase[3];
            const bool fits = s_cnt + tot <= RLIST;                  // uniform
            if (fits && hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(i - c0);
            __syncthreads();
            if (!fits) break;                                        // this chunk is redone after the flush
            if (tid == 0) s_cnt += tot;
            __syncthreads();
        }
        const int nl = s_cnt;
        // ---- paint: every thread walks the list for its own pixels ------------------------------------------------
        for (int e = 0; e < nl; ++e) {
            const av_prim q = pr[c0 + list[e]];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (y < H && xb + k < W && prim_covers(q, xb + k, y, vt)) px[k] = prim_apply(q, px[k]);
        }
        __syncthreads();
        c0 = c;                                                      // first primitive not yet listed
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (y < H && xb + k < W) {
            uint8_t* p = base + ((size_t)y * W + xb + k) * 3;
            p[0] = (uint8_t)(px[k] & 255u), p[1] = (uint8_t)((px[k] >> 8) & 255u), p[2] = (uint8_t)((px[k] >> 16) & 255u);
        }
}

// bilinear resize (half-pixel centres, replicate border), u8 BGR: create_side_by_side's cv2.resize (overlays.py:187-193)
__global__ void resize_kernel(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw, int dpitch,
                              int dx0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dh * dw) return;
    const int y = i / dw, x = i - y * dw;
    // source position in 1/65536 pixels: (x + 0.5) * sw / dw - 0.5
    const long long fx = ((2LL * x + 1) * sw * 32768LL) / dw - 32768LL, fy = ((2LL * y + 1) * sh * 32768LL) / dh - 32768LL;
    const long long cx = fx < 0 ? 0 : fx, cy = fy < 0 ? 0 : fy;
    int x0 = (int)(cx >> 16), y0 = (int)(cy >> 16);
    const int wx = (int)(cx & 65535), wy = (int)(cy & 65535);
    const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    x0 = min(x0, sw - 1), y0 = min(y0, sh - 1);
    uint8_t* o = dst + ((size_t)y * dpitch + dx0 + x) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long p00 = src[((size_t)y0 * sw + x0) * 3 + k], p01 = src[((size_t)y0 * sw + x1) * 3 + k];
        const long long p10 = src[((size_t)y1 * sw + x0) * 3 + k], p11 = src[((size_t)y1 * sw + x1) * 3 + k];
        const long long top = p00 * (65536 - wx) + p01 * wx, bot = p10 * (65536 - wx) + p11 * wx;
        o[k] = (uint8_t)((top * (65536 - wy) + bot * wy + (1LL << 31)) >> 32);
    }
}


// ---- BEV panel straight from the hot loop's tables (no host round trip) ------------------------------------------------
// One workgroup per stream writes the panel's primitive list in BEVRenderer.render's order (bev_renderer.py:304-348) into
// fixed slot ranges (unused slots stay type 0): candidates of rank 1.. in grey, the planned path (rank 0) in green with
// its waypoint discs, the confirmed tracks (footprint, outline, heading arrow, "ID:n", trail), the ego vehicle with its
// uncertainty circle, the legend.  Inputs are what av_planner_plan / av_tracker_update / av_kf_step left in HBM.
struct BevGeom {
    double x_lo, y_lo, xs, ys, ppm;
    int W, H;
};
__device__ __forceinline__ void w2p(const BevGeom& g, double x, double y, int& px, int& py) {
    px = (int)((x - g.x_lo) * g.xs), py = (int)((double)g.H - (y - g.y_lo) * g.ys);          // bev_renderer.py:70-83 (int() truncates)
}
__device__ __forceinline__ av_prim mk(int type, int x0, int y0, int x1, int y1, int p, unsigned bgr) {
    av_prim q{};
    q.type = type, q.x0 = x0, q.y0 = y0, q.x1 = x1, q.y1 = y1, q.p = p;
    q.b = (uint8_t)(bgr & 255u), q.g = (uint8_t)((bgr >> 8) & 255u), q.r = (uint8_t)(bgr >> 16);
    return q;
}
__device__ __forceinline__ int cv_round_d(double v) { return (int)rint(v); }
// footprint + outline + arrow of one vehicle: 8 primitives at out[0..8)
__device__ void vehicle_prims(const BevGeom& g, av_prim* out, double x, double y, double heading, double length, double width, unsigned bgr) {
    const double ch = cos(heading), sh = sin(heading), hl = length / 2, hw = width / 2;
    const double cxs[4] = {x + hl * ch - hw * sh, x + hl * ch + hw * sh, x - hl * ch + hw * sh, x - hl * ch - hw * sh};
    const double cys[4] = {y + hl * sh + hw * ch, y + hl * sh - hw * ch, y - hl * sh - hw * ch, y - hl * sh + hw * ch};
    int px[4], py[4];
    for (int k = 0; k < 4; ++k) w2p(g, cxs[k], cys[k], px[k], py[k]);
    av_prim q = mk(AV_PRIM_QUAD, px[0], py[0], px[1], py[1], 0, bgr);
    q.x2 = px[2], q.y2 = py[2], q.x3 = px[3], q.y3 = py[3];
    out[0] = q;
    for (int k = 0; k < 4; ++k) out[1 + k] = mk(AV_PRIM_SEG, px[k], py[k], px[(k + 1) & 3], py[(k + 1) & 3], 1, 0xFFFFFFu);
    int fx, fy, cx, cy;
    w2p(g, x + hl * ch, y + hl * sh, fx, fy);
    w2p(g, x, y, cx, cy);
    out[5] = mk(AV_PRIM_SEG, cx, cy, fx, fy, 2, 0xFFFFFFu);                     // cv2.arrowedLine(center, front, tipLength=0.5)
    const double tip = sqrt((double)((cx - fx) * (cx - fx) + (cy - fy) * (cy - fy))) * 0.5, ang = atan2((double)(cy - fy), (double)(cx - fx));
    for (int k = 0; k < 2; ++k) {
        const double a = ang + (k ? -0.78539816339744830962 : 0.78539816339744830962);
        out[6 + k] = mk(AV_PRIM_SEG, cv_round_d(fx + tip * cos(a)), cv_round_d(fy + tip * sin(a)), fx, fy, 2, 0xFFFFFFu);
    }
}
__device__ void text_prims(av_prim* out, int cap, const char* s, int n, int x, int y_base, unsigned bgr) {
    for (int k = 0; k < cap; ++k)
        out[k] = k < n && s[k] > 32 ? mk(AV_PRIM_GLYPH, x + 6 * k, y_base - 7, 1, 0, (int)s[k], bgr) : av_prim{};
}

constexpr int BV_AGENT = 72, BV_LABEL = 12, BV_EGO = 12, BV_LEGEND = 19;

__global__ void __launch_bounds__(256) bev_build_kernel(int n_streams, int n_frames, int frame, av_bev_cfg cfg, int tcap, int L,
                                                        const av_track_row* __restrict__ snap, const int32_t* __restrict__ snap_n,
                                                        const uint8_t* __restrict__ trk_state, size_t trk_bytes,
                                                        const double* __restrict__ vstate, const double* __restrict__ wp,
                                                        const int32_t* __restrict__ order, int n_cand, int n_pts, int prim_cap,
                                                        av_prim* __restrict__ prims, int32_t* __restrict__ n_prims) {
    const int s = blockIdx.x, tid = threadIdx.x;
    BevGeom g;
    g.x_lo = cfg.x_min, g.y_lo = cfg.y_min, g.W = cfg.width, g.H = cfg.height, g.ppm = cfg.pixels_per_meter;
    g.xs = (double)cfg.width / (cfg.x_max - cfg.x_min), g.ys = (double)cfg.height / (cfg.y_max - cfg.y_min);
    av_prim* out = prims + (size_t)s * prim_cap;
    const size_t sf = (size_t)s * n_frames + frame;
    const int nd = min(cfg.n_candidates, n_cand), segs = n_pts - 1, ndisc = (n_pts + 2) / 3;
    // ---- trajectories: rank r of the stable order; rank 0 is the planned one and is drawn last -----------------
    const double* wps = wp + sf * n_cand * n_pts * 6;
    const int32_t* ord = order + sf * n_cand;
    int base = 0;
    for (int i = tid; i < (nd > 0 ? nd - 1 : 0) * segs; i += 256) {
        const int r = 1 + i / segs, k = i % segs;
        const double* a = wps + ((size_t)ord[r] * n_pts + k) * 6;
        int x0, y0, x1, y1;
        w2p(g, a[0], a[1], x0, y0), w2p(g, a[6], a[7], x1, y1);
        out[base + i] = mk(AV_PRIM_SEG, x0, y0, x1, y1, 1, 0x505050u);
    }
    base += (cfg.n_candidates > 0 ? cfg.n_candidates - 1 : 0) * segs;
    for (int i = tid; i < (cfg.n_candidates - 1 - (nd > 0 ? nd - 1 : 0)) * segs; i += 256) out[(nd > 0 ? nd - 1 : 0) * segs + i] = av_prim{};
    if (nd > 0) {
        const double* pw = wps + (size_t)ord[0] * n_pts * 6;
        for (int k = tid; k < segs; k += 256) {
            int x0, y0, x1, y1;
            w2p(g, pw[k * 6], pw[k * 6 + 1], x0, y0), w2p(g, pw[k * 6 + 6], pw[k * 6 + 7], x1, y1);
            out[base + k] = mk(AV_PRIM_SEG, x0, y0, x1, y1, 3, 0x00FF00u);
        }
        for (int k = tid; k < ndisc; k += 256) {
            int x0, y0;
            w2p(g, pw[3 * k * 6], pw[3 * k * 6 + 1], x0, y0);
            out[base + segs + k] = mk(AV_PRIM_DISC, x0, y0, 0, 0, 3, 0x00FF00u);
        }
    } else {
        for (int k = tid; k < segs + ndisc; k += 256) out[base + k] = av_prim{};
    }
    base += segs + ndisc;
    // ---- agents: confirmed rows in table order (multi_object_tracker.py:236-241 -> bev_renderer.py:185-235) ----------
    __shared__ int rank[1024];
    const av_track_row* rows = snap + sf * tcap;
    const int nrows = min(snap_n[sf], tcap);
    for (int i = tid; i < tcap; i += 256) rank[i] = (i < nrows && (rows[i].flags & 1)) ? 1 : 0;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < tcap; ++i) {
            const int v = rank[i];
            rank[i] = v ? acc : -1;
            acc += v;
        }
    }
    __syncthreads();
    const unsigned agent_cols[6] = {0x00FF00u, 0x0000FFu, 0xFF0000u, 0x00FFFFu, 0xFF00FFu, 0xFFFF00u};     // BGR packed b | g<<8 | r<<16
    const double* hist = reinterpret_cast<const double*>(trk_state + (size_t)s * trk_bytes + 64 + (size_t)tcap * 64);
    for (int i = tid; i < tcap * BV_AGENT; i += 256) out[base + i] = av_prim{};
    __syncthreads();
    for (int i = tid; i < tcap; i += 256) {
        if (rank[i] < 0) continue;
        const av_track_row r = rows[i];
        av_prim* o = out + base + rank[i] * BV_AGENT;
        const unsigned col = agent_cols[((r.id % 6) + 6) % 6];
        const double cx = (double)(r.x1 + r.x2) / 2.0, cy = (double)(r.y1 + r.y2) / 2.0;
        const double wx = (cx - 320.0) * 0.03, wy = 50.0 - cy * 0.1;
        vehicle_prims(g, o, wx, wy, 0.0, 3.0, 1.5, col);
        char lab[BV_LABEL] = {'I', 'D', ':'};
        int n = 3, id = r.id, dig[10], nd2 = 0;
        do { dig[nd2++] = id % 10, id /= 10; } while (id > 0 && nd2 < 9);
        while (nd2 > 0 && n < BV_LABEL) lab[n++] = (char)('0' + dig[--nd2]);
        int pcx, pcy;
        w2p(g, wx, wy, pcx, pcy);
        text_prims(o + 8, BV_LABEL, lab, n, pcx - 20, pcy - 15, 0xFFFFFFu);
        // trail: the ring's entries max(0, hist_len - L) .. hist_len - 1, segment j with thickness max(1, int(2 j / len))
        const int hl = r.hist_len, len = hl < L ? hl : L, first = hl - len;
        av_prim* tr = o + 8 + BV_LABEL;
        for (int j = 1; j < len && j - 1 < BV_AGENT - 8 - BV_LABEL; ++j) {
            const double* a = hist + ((size_t)r.slot * L + (size_t)((first + j - 1) % L)) * 4;
            const double* b = hist + ((size_t)r.slot * L + (size_t)((first + j) % L)) * 4;
            int x0, y0, x1, y1;
            w2p(g, (a[0] - 320.0) * 0.03, 50.0 - a[1] * 0.1, x0, y0), w2p(g, (b[0] - 320.0) * 0.03, 50.0 - b[1] * 0.1, x1, y1);
            const int th = (int)(2.0 * ((double)j / (double)len));
            tr[j - 1] = mk(AV_PRIM_SEG, x0, y0, x1, y1, th > 1 ? th : 1, col);
        }
    }
    base += tcap * BV_AGENT;
    // ---- ego vehicle, its uncertainty circle, the legend ---------------------------------------------------------------
    if (tid == 0) {
        av_prim* o = out + base;
        for (int k = 0; k < BV_EGO + BV_LEGEND; ++k) o[k] = av_prim{};
        if (vstate) {
            const double* v = vstate + sf * AV_VSTATE_DOUBLES;
            vehicle_prims(g, o, v[0], v[1], v[4], 4.5, 2.0, 0xFFC800u);           // ego colour (0, 200, 255) in BGR
            int cx, cy;
            w2p(g, v[0], v[1], cx, cy);
            text_prims(o + 8, 3, "EGO", 3, cx - 20, cy - 15, 0xFFFFFFu);
            const int rad = (int)(v[9] * g.ppm);
            if (rad > 0) o[11] = mk(AV_PRIM_RING, cx, cy, 0, 0, rad, 0xFFFF00u);   // (0, 255, 255) in BGR
        }
        av_prim* lg = o + BV_EGO;
        const char* names[3] = {"EGO", "Planned", "Agents"};
        const int lens[3] = {3, 7, 6};
        const unsigned cols[3] = {0xFFC800u, 0x00FF00u, 0x00FF00u};
        int y = 20, at = 0;
        for (int k = 0; k < 3; ++k) {
            lg[at++] = mk(AV_PRIM_RECT, 10, y - 10, 25, y + 5, 0, cols[k]);
            text_prims(lg + at, lens[k], names[k], lens[k], 30, y, 0xFFFFFFu);
            at += lens[k];
            y += 20;
        }
        n_prims[s] = base + BV_EGO + BV_LEGEND;
    }
}

// f-4: planar YUV 4:2:0 (I420) -> interleaved BGR, the pixel work of video ingest (what a decoder hands
// over; cv2.VideoCapture.read returns BGR, video_loader.py:96-106).  ITU-R BT.601 limited range in OpenCV's 20-bit fixed
// point (cvtColor COLOR_YUV2BGR_I420: CY 1220542, CUB 2116026, CUG -409993, CVG -852492, CVR 1673527), restated from its
// published source -- parity unpinned.  A thread makes two horizontally adjacent pixels (they share a chroma sample).
__global__ void i420_to_bgr_kernel(const uint8_t* __restrict__ yuv, int n, int h, int w, uint8_t* __restrict__ bgr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, w2 = w >> 1;
    if (i >= n * h * w2) return;
    const int f = i / (h * w2), r = i - f * h * w2, y = r / w2, x = (r - y * w2) * 2;
    const size_t fsz = (size_t)h * w * 3 / 2;
    const uint8_t* Y = yuv + (size_t)f * fsz;
    const uint8_t* U = Y + (size_t)h * w;
    const uint8_t* V = U + (size_t)(h >> 1) * w2;
    const int u = (int)U[(size_t)(y >> 1) * w2 + (x >> 1)] - 128, v = (int)V[(size_t)(y >> 1) * w2 + (x >> 1)] - 128;
    const int ruv = (1 << 19) + 1673527 * v, guv = (1 << 19) - 852492 * v - 409993 * u, buv = (1 << 19) + 2116026 * u;
    uint8_t* o = bgr + ((size_t)f * h * w + (size_t)y * w + x) * 3;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int yy = max(0, (int)Y[(size_t)y * w + x + k] - 16) * 1220542;
        const int b = (yy + buv) >> 20, g = (yy + guv) >> 20, rr = (yy + ruv) >> 20;
        o[3 * k] = (uint8_t)min(max(b, 0), 255), o[3 * k + 1] = (uint8_t)min(max(g, 0), 255), o[3 * k + 2] = (uint8_t)min(max(rr, 0), 255);
    }
}

}  // namespace

extern "C" {

int av_raster_draw(av_ctx* ctx, av_stream_t stream, int n_images, int h, int w, uint8_t* img, const av_prim* prims, int prim_cap,
                   const int32_t* n_prims, const int32_t* verts, int vert_cap) {
    AV_REQUIRE(ctx && img && prims && n_prims, AV_EINVAL, "av_raster_draw: null argument");
    AV_REQUIRE(n_images > 0 && h > 0 && w > 0 && h < 8192 && w < 8192, AV_EINVAL, "av_raster_draw: bad image size %dx%d", w, h);
    AV_REQUIRE(prim_cap > 0 && prim_cap <= 65535 && vert_cap >= 0, AV_EINVAL, "av_raster_draw: prim_cap must be in [1, 65535]");
    static_assert(sizeof(av_prim) == 48, "av_prim layout");
    const dim3 grid((w + RT - 1) / RT, (h + RT - 1) / RT, n_images);
    hipLaunchKernelGGL(raster_kernel, grid, dim3(256), 0, as_stream(stream), n_images, h, w, img, prims, prim_cap, n_prims, verts, vert_cap);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_bev_prim_cap(const av_bev_cfg* cfg, int tcap, int n_points) {
    if (!cfg || tcap <= 0 || n_points < 2) return 0;
    const int segs = n_points - 1, ndisc = (n_points + 2) / 3;
    return (cfg->n_candidates > 0 ? cfg->n_candidates - 1 : 0) * segs + segs + ndisc + tcap * BV_AGENT + BV_EGO + BV_LEGEND;
}

int av_bev_build(av_ctx* ctx, av_stream_t stream, const av_bev_cfg* cfg, int n_streams, int n_frames, int frame, int tcap,
                 int trajectory_length, const av_track_row* snap, const int32_t* snap_n, const void* tracker_state,
                 const double* vstate, const double* waypoints, const int32_t* order, av_prim* prims, int prim_cap, int32_t* n_prims) {
    AV_REQUIRE(ctx && cfg && snap && snap_n && tracker_state && waypoints && order && prims && n_prims, AV_EINVAL,
               "av_bev_build: null argument");
    AV_REQUIRE(ctx->planner_ready, AV_ESTATE, "av_bev_build: the planner's dimensions are not configured");
    AV_REQUIRE(n_streams > 0 && n_frames > 0 && frame >= 0 && frame < n_frames && tcap > 0 && tcap <= 1024 && trajectory_length > 0, AV_EINVAL,
               "av_bev_build: bad dimensions");
    AV_REQUIRE(cfg->width > 0 && cfg->height > 0 && cfg->x_max > cfg->x_min && cfg->y_max > cfg->y_min && cfg->n_candidates >= 0, AV_EINVAL,
               "av_bev_build: bad panel configuration");
    const int need = av_bev_prim_cap(cfg, tcap, ctx->n_points);
    AV_REQUIRE(prim_cap >= need && prim_cap <= 65535, AV_EINVAL, "av_bev_build: prim_cap %d, need %d (<= 65535)", prim_cap, need);
    hipLaunchKernelGGL(bev_build_kernel, dim3(n_streams), dim3(256), 0, as_stream(stream), n_streams, n_frames, frame, *cfg, tcap,
                       trajectory_length, snap, snap_n, (const uint8_t*)tracker_state, av_tracker_state_bytes(tcap, trajectory_length),
                       vstate, waypoints, order, ctx->n_cand, ctx->n_points, prim_cap, prims, n_prims);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_i420_to_bgr(av_ctx* ctx, av_stream_t stream, int n_frames, int h, int w, const uint8_t* yuv, uint8_t* bgr) {
    AV_REQUIRE(ctx && yuv && bgr, AV_EINVAL, "av_i420_to_bgr: null argument");
    AV_REQUIRE(n_frames > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, AV_EINVAL, "av_i420_to_bgr: 4:2:0 frames need even sizes, got %dx%d", w, h);
    const long n = (long)n_frames * h * (w / 2);
    hipLaunchKernelGGL(i420_to_bgr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), yuv, n_frames, h, w, bgr);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_resize_into(av_ctx* ctx, av_stream_t stream, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, int dst_pitch_px,
                   int dst_x0) {
    AV_REQUIRE(ctx && src && dst, AV_EINVAL, "av_resize_into: null argument");
    AV_REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && dst_x0 >= 0 && dst_x0 + dw <= dst_pitch_px, AV_EINVAL, "av_resize_into: bad geometry");
    const int n = dh * dw;
    hipLaunchKernelGGL(resize_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), src, sh, sw, dst, dh, dw, dst_pitch_px, dst_x0);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"
