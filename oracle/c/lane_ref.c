/*
 * lane_ref.c -- CPU oracle for the lane pixel path (TEST INFRASTRUCTURE, not product code).
 *
 * PARITY UNPINNED: OpenCV (requirements.txt:2, opencv-python >= 4.5.0, no lock) is not installed in the
 * build container and the reference ships no tests, so nothing pins these functions to the real library.
 * They restate, from the published OpenCV 4.x algorithms, the calls LaneDetector makes
 * (src/perception/lane_detector.py):
 *   cvtColor(BGR2GRAY)            :69   gray = (1868 B + 9617 G + 4899 R + 8192) >> 14
 *   GaussianBlur((5,5), 0)        :72   fixed-point separable [1 4 6 4 1]/16, BORDER_REFLECT_101,
 *                                       result (sum of w_i w_j p + 128) >> 8
 *   np.median                     :79   mean of the two middle order statistics (even count)
 *   Canny(blur, lo, hi)           :83   3x3 Sobel (BORDER_REPLICATE), |gx|+|gy|, sector NMS with
 *                                       tan(22.5 deg) = 13573/2^15, 8-connected hysteresis
 *   fillPoly + bitwise_and        :63,89 trapezoid ROI
 *   HoughLinesP(1, pi/180, 50, minLineLength=50, maxLineGap=150)  :94-101
 *                                       progressive probabilistic Hough: fixed-seed cv::RNG, random
 *                                       pixel order, vote, walk along the strongest line, un-vote
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
static inline int clampi(int p, int n) { return p < 0 ? 0 : (p >= n ? n - 1 : p); }

void lane_gray(const uint8_t* bgr, int h, int w, uint8_t* gray) {
    for (long i = 0; i < (long)h * w; ++i)
        gray[i] = (uint8_t)((1868 * bgr[3 * i] + 9617 * bgr[3 * i + 1] + 4899 * bgr[3 * i + 2] + 8192) >> 14);
}

void lane_blur5(const uint8_t* gray, int h, int w, uint8_t* out) {
    static const int k[5] = {1, 4, 6, 4, 1};
    int* tmp = (int*)malloc(sizeof(int) * (size_t)h * w);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int d = -2; d <= 2; ++d) s += k[d + 2] * gray[(long)y * w + reflect101(x + d, w)];
            tmp[(long)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int d = -2; d <= 2; ++d) s += k[d + 2] * tmp[(long)reflect101(y + d, h) * w + x];
            out[(long)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    free(tmp);
}

/* np.median of a uint8 image -> float64; thresholds as lane_detector.py:79-81 */
double lane_median(const uint8_t* img, long n, int* lo, int* hi) {
    long hist[256] = {0};
    for (long i = 0; i < n; ++i) hist[img[i]]++;
    long k1 = (n - 1) / 2, k2 = n / 2, c = 0;
    int v1 = -1, v2 = -1;
    for (int v = 0; v < 256; ++v) {
        c += hist[v];
        if (v1 < 0 && c > k1) v1 = v;
        if (v2 < 0 && c > k2) { v2 = v; break; }
    }
    double med = ((double)v1 + (double)v2) / 2.0;
    double l = 0.7 * med, u = 1.3 * med;
    *lo = (int)(l > 0 ? l : 0);
    *hi = (int)(u < 255 ? u : 255);
    return med;
}

/* map: 0 = weak candidate, 1 = not an edge, 2 = strong.  edges: 255 where connected to a strong pixel. */
void lane_canny(const uint8_t* img, int h, int w, int lo, int hi, uint8_t* edges) {
    if (lo > hi) { int t = lo; lo = hi; hi = t; }
    long n = (long)h * w;
    int16_t* dx = (int16_t*)malloc(sizeof(int16_t) * n);
    int16_t* dy = (int16_t*)malloc(sizeof(int16_t) * n);
    int* mag = (int*)calloc((size_t)(h + 2) * (w + 2), sizeof(int));   /* zero border */
    uint8_t* map = (uint8_t*)malloc((size_t)n);
#define P(yy, xx) ((int)img[(long)clampi(yy, h) * w + clampi(xx, w)])
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int gx = (P(y - 1, x + 1) + 2 * P(y, x + 1) + P(y + 1, x + 1)) - (P(y - 1, x - 1) + 2 * P(y, x - 1) + P(y + 1, x - 1));
            int gy = (P(y + 1, x - 1) + 2 * P(y + 1, x) + P(y + 1, x + 1)) - (P(y - 1, x - 1) + 2 * P(y - 1, x) + P(y - 1, x + 1));
            dx[(long)y * w + x] = (int16_t)gx;
            dy[(long)y * w + x] = (int16_t)gy;
            mag[(long)(y + 1) * (w + 2) + x + 1] = abs(gx) + abs(gy);
        }
#undef P
#define M(yy, xx) mag[(long)((yy) + 1) * (w + 2) + (xx) + 1]
    const int TG22 = 13573;
    long* stack = (long*)malloc(sizeof(long) * n);
    long sp = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            long i = (long)y * w + x;
            int m = M(y, x), v = 1;
            if (m > lo) {
                int xs = dx[i], ys = dy[i];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                int is_max;
                if (ay < tg22x) is_max = m > M(y, x - 1) && m >= M(y, x + 1);
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) is_max = m > M(y - 1, x) && m >= M(y + 1, x);
                    else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        is_max = m > M(y - 1, x - s) && m > M(y + 1, x + s);
                    }
                }
                if (is_max) v = m > hi ? 2 : 0;
            }
            map[i] = (uint8_t)v;
            if (v == 2) stack[sp++] = i;
        }
#undef M
    while (sp > 0) {
        long i = stack[--sp];
        int y = (int)(i / w), x = (int)(i % w);
        for (int d = 0; d < 8; ++d) {
            static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            int yy = y + oy[d], xx = x + ox[d];
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            long j = (long)yy * w + xx;
            if (map[j] == 0) { map[j] = 2; stack[sp++] = j; }
        }
    }
    for (long i = 0; i < n; ++i) edges[i] = map[i] == 2 ? 255 : 0;
    free(dx); free(dy); free(mag); free(map); free(stack);
}

/* Trapezoid ROI of lane_detector.py:55-60: (x0,h) (x1,yt) (x2,yt) (x3,h).  Row y in [yt, h) keeps
 * columns xl(y)..xr(y), the edge abscissae rounded to nearest (ties up) in exact integer arithmetic. */
void lane_roi_bounds(int h, int w, int y, int* xl, int* xr) {
    int x0 = (int)(w * 0.1), x1 = (int)(w * 0.4), x2 = (int)(w * 0.6), x3 = (int)(w * 0.9), yt = (int)(h * 0.6);
    if (y < yt || y >= h) { *xl = 1; *xr = 0; return; }
    long den = h - yt, t = h - y;                 /* 0 at the bottom vertex row (outside), den at the top */
    *xl = (int)((2 * (x0 * den + (long)(x1 - x0) * t) + den) / (2 * den));
    *xr = (int)((2 * (x3 * den + (long)(x2 - x3) * t) + den) / (2 * den));
}
void lane_apply_roi(uint8_t* edges, int h, int w) {
    for (int y = 0; y < h; ++y) {
        int xl, xr;
        lane_roi_bounds(h, w, y, &xl, &xr);
        for (int x = 0; x < w; ++x)
            if (x < xl || x > xr) edges[(long)y * w + x] = 0;
    }
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }     /* round half to even, like cvRound */

/* cv::HoughLinesProbabilistic.  lines: [max_lines][4] (x1,y1,x2,y2).  Returns the number of lines. */
int lane_houghp(const uint8_t* image, int h, int w, int threshold, int line_length, int line_gap, int max_lines,
                int32_t* lines) {
    const float rho = 1.0f, theta = (float)(3.14159265358979323846 / 180.0);
    const int numangle = (int)lrint(3.14159265358979323846 / theta);
    const int numrho = (int)lrint(((w + h) * 2 + 1) / rho);
    const float irho = 1 / rho;
    int* accum = (int*)calloc((size_t)numangle * numrho, sizeof(int));
    uint8_t* mask = (uint8_t*)malloc((size_t)h * w);
    float* tt = (float*)malloc(sizeof(float) * 2 * numangle);
    int32_t* nz = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)h * w);
    for (int n = 0; n < numangle; ++n) {
        tt[2 * n] = (float)(cos((double)n * theta) * irho);
        tt[2 * n + 1] = (float)(sin((double)n * theta) * irho);
    }
    long count = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int on = image[(long)y * w + x] != 0;
            mask[(long)y * w + x] = (uint8_t)on;
            if (on) { nz[2 * count] = x; nz[2 * count + 1] = y; ++count; }
        }
    uint64_t state = (uint64_t)-1;
    int nlines = 0;
    const int shift = 16;
    for (; count > 0; --count) {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        long idx = (long)((unsigned)state % (unsigned)count);
        int j = nz[2 * idx], i = nz[2 * idx + 1];
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(long)i * w + j]) continue;
        int max_val = threshold - 1, max_n = 0;
        for (int n = 0; n < numangle; ++n) {
            int r = cv_round_f(j * tt[2 * n] + i * tt[2 * n + 1]) + (numrho - 1) / 2;
            int val = ++accum[(long)n * numrho + r];
            if (max_val < val) { max_val = val; max_n = n; }
        }
        if (max_val < threshold) continue;
        float a = -tt[2 * max_n + 1], b = tt[2 * max_n];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = cv_round_f(b * (1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = cv_round_f(a * (1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        int ex[2] = {0, 0}, ey[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) { dx = -dx; dy = -dy; }
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; } else { j1 = x >> shift; i1 = y; }
                if (j1 < 0 || j1 >= w || i1 < 0 || i1 >= h) break;
                if (mask[(long)i1 * w + j1]) { gap = 0; ey[k] = i1; ex[k] = j1; }
                else if (++gap > line_gap) break;
            }
        }
        int good = abs(ex[1] - ex[0]) >= line_length || abs(ey[1] - ey[0]) >= line_length;
        for (int k = 0; k < 2; ++k) {
            int x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) { dx = -dx; dy = -dy; }
            for (;; x += dx, y += dy) {
                int i1, j1;
                if (xflag) { j1 = x; i1 = y >> shift; } else { j1 = x >> shift; i1 = y; }
                uint8_t* m = mask + (long)i1 * w + j1;
                if (*m) {
                    if (good)
                        for (int n = 0; n < numangle; ++n) {
                            int r = cv_round_f(j1 * tt[2 * n] + i1 * tt[2 * n + 1]) + (numrho - 1) / 2;
                            accum[(long)n * numrho + r]--;
                        }
                    *m = 0;
                }
                if (i1 == ey[k] && j1 == ex[k]) break;
            }
        }
        if (good) {
            lines[4 * nlines] = ex[0]; lines[4 * nlines + 1] = ey[0];
            lines[4 * nlines + 2] = ex[1]; lines[4 * nlines + 3] = ey[1];
            if (++nlines >= max_lines) break;
        }
    }
    free(accum); free(mask); free(tt); free(nz);
    return nlines;
}
