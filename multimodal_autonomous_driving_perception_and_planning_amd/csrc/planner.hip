// P1-P3: batched candidate-trajectory planner.
//
// Reference: MotionPlanner (src/planning/motion_planner.py)
//   generate_polynomial_trajectory :126-204, evaluate_trajectory_cost :206-262, plan :264-303.
//
// Data layout (HBM):  waypoints [n_states][C][n][6] float64, AoS per waypoint
//   (x, y, heading, velocity, timestamp, curvature) -- the field order of the Waypoint dataclass --
//   so one trajectory is one contiguous n*48-byte run; cost/order [n_states][C].
//
// Mapping: a 256-thread workgroup owns G consecutive start states.
//   phase 1  3*G lanes run the sequential part the reference has per speed option: the velocity
//            blend, the prefix sum s_i = s_{i-1} + v_i*dt (:156-157) and the velocity/acceleration
//            cost accumulated left to right (:235-244).  These depend only on (v0, vt) and are
//            shared by the num_samples lateral offsets.  G more lanes take cos/sin of the heading.
//   phase 2  each wave takes whole trajectories; lane = waypoint.  Positions, tangent heading
//            (atan2 of the forward difference), curvature and the per-waypoint cost terms are
//            computed in registers, the AoS image of the trajectory is assembled in a per-wave LDS
//            tile, and the tile is streamed out as contiguous 16-byte-per-lane stores (the kernel's
//            HBM traffic is these stores: 48 B per waypoint).
//   phase 3  stable rank of the C costs (== Python's stable sort, :300).
// All arithmetic is float64 in the reference's operation order; the library is built with
// -ffp-contract=off so a*b+c stays two roundings like NumPy scalar code.
#include "common.h"

#include <cmath>



namespace {

struct PlanParams {
    int n, n_lat, C;
    double dt, H;
    double w_lat, w_vel, w_acc, w_curv;
    const double *t, *alpha, *q, *dtd, *lat;
    double atq[20];      // atan polynomial (kernel argument -> scalar registers, one per Horner step)
};

// atan(t) = t + t*z*Q(z), z = t*t, t in [0,1]: degree-19 Chebyshev interpolant of (atan(sqrt z)/sqrt z - 1)/z
// derived in long double by tools/atan_poly.py; float64 Horner error <= 2.1 ulp over [0,1].
static const double ATAN_Q[20] = {
    -0x1.5555555555549p-2, 0x1.99999999979b9p-3,  -0x1.24924923e22b4p-3, 0x1.c71c718e5a102p-4,
    -0x1.745d120df88fbp-4, 0x1.3b1362fc35ab5p-4,  -0x1.110deef367776p-4, 0x1.e1b3c8d50e91fp-5,
    -0x1.ae2c7119eb922p-5, 0x1.81fdd2525246ap-5,  -0x1.56daf4786fd80p-5, 0x1.2575526b2309ap-5,
    -0x1.d168e01adf666p-6, 0x1.462cc26bda000p-6,  -0x1.8055065b73333p-7, 0x1.693bd6b79999ap-8,
    -0x1.fdf4e15cccccdp-10, 0x1.f224118000000p-12, -0x1.2568700000000p-14, 0x1.2ed799999999ap-18,
};

// atan2 for finite arguments of ordinary magnitude (waypoint deltas), |error| <= ~2 ulp like libm's.
// NumPy's arctan2 is libm's, so this call is tolerance-matched either way; what this version avoids is
// ocml's special-case handling and the ~40 moves that materialise its 64-bit constants per call.
__device__ __forceinline__ double atan2_fast(double y, double x, const double (&Q)[20]) {
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const bool swap = ay > ax;
    const double mx = swap ? ay : ax, mn = swap ? ax : ay;
    // t = mn / mx: reciprocal refined twice (as the compiler's own f64 divide does), then one residual
    // correction of the quotient -> correctly rounded but for astronomically rare operands
    double r = __builtin_amdgcn_rcp(mx);
    r = __builtin_fma(__builtin_fma(-mx, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-mx, r, 1.0), r, r);
    double t = mn * r;
    t = __builtin_fma(__builtin_fma(-t, mx, mn), r, t);
    t = mx == 0.0 ? 0.0 : t;                                     // atan2(0, 0) = 0
    const double z = t * t;
    double pz = Q[19];
#pragma unroll
    for (int k = 18; k >= 0; --k) pz = __builtin_fma(pz, z, Q[k]);
    double a = __builtin_fma(t * z, pz, t);
    a = swap ? 1.5707963267948966 - a : a;
    a = x < 0.0 ? 3.141592653589793 - a : a;
    return __builtin_copysign(a, y);
}

__host__ __device__ inline int even_up(int v) { return (v + 1) & ~1; }

// doubles of dynamic LDS for G start states per workgroup of NW waves
__host__ __device__ inline size_t plan_lds_doubles(int G, int n, int C, int NW = 4) {
    return (size_t)G * 3 * n * 2 + even_up(G * 3 * 3) + (size_t)G * 8 + (size_t)3 * even_up(G * C) + (size_t)NW * n * 6;
}

// Orders this wave's LDS accesses for the compiler.  The hardware executes one wave's DS operations
// in issue order, so no s_waitcnt is needed (and none for outstanding global stores either).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// The workgroup-cooperative planner for small batches (and n > 64): G consecutive start states f0 .. f0 + G - 1 by a workgroup of
// NW waves.  A device function, so that the fused time-step kernel (step.hip) runs the very same code after its Kalman step.
//   phase 1  per (state, speed) pair, one wave each: lanes make the per-waypoint terms in parallel (velocity blend v_i, v_i dt, the
//            velocity and acceleration cost terms -- the 50 float64 divides of the acceleration used to sit in ONE lane's
//            sequential loop) and the wave adds up the arc length s_i = s_{i-1} + v_i dt left to right, the reference's order
//            (:156-157), terms broadcast from registers by v_readlane: no memory access on the sequential path.  Only this
//            chain is in front of the barrier; the two cost chains (:235-244) have no consumer before phase 3 and run behind
//            it, in the same waves, which in exchange get fewer trajectories (in-kernel cycle stamps: the three chains back to
//            back were 6 of the kernel's 14 us for one start state)
//   phase 2  each wave takes whole trajectories; lane = waypoint (positions, tangent heading by atan2_fast, curvature, cost terms;
//            AoS image assembled in a per-wave LDS tile and streamed out as 16-byte-per-lane stores); per-trajectory sums by DPP
//   phase 3  costs assembled in the reference's order of accumulation, stable rank of the C costs (== Python's stable sort, :300)
template <int G, int NW>
__device__ __forceinline__ void plan_block(const PlanParams& p, int f0, int n_states, const double* __restrict__ state,
                                           const double* __restrict__ ref, int n_ref, const double* __restrict__ obs, int n_obs,
                                           double* __restrict__ wp, double* __restrict__ cost, int32_t* __restrict__ order, double* sm) {
    const int n = p.n, C = p.C;
    double* vs = sm;                                   // [G][3][n][2]  (v, s)
    double* base = vs + (size_t)G * 3 * n * 2;         // [G][3][3]     S_v, S_a, running(S_v then acc terms)
    double* trig = base + even_up(G * 3 * 3);          // [G][8]        x0 y0 cos sin cos(h+pi/2) sin(h+pi/2) h0
    double* costs = trig + G * 8;                      // [3][G][C]     per trajectory: reference-path, curvature, obstacle sums; then [0]: the cost
    double* stage_all = costs + 3 * even_up(G * C);    // [NW][n*6]
    const int CS = even_up(G * C);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double* stage = stage_all + (size_t)wid * n * 6;
    constexpr int P = G * 3;                           // (state, speed) pairs
    static_assert(P <= NW || NW == 4, "one wave per pair keeps the cost terms in its registers");

    auto bcast = [&](double x, int i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(x), i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)__double_as_longlong(x) >> 32), i);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    auto terms = [&](double v0, double dv, int i, double& v, double& vdt, double& velt, double& acct) {
        v = v0 + dv * p.alpha[i];                             // :153-154
        vdt = v * p.dt;
        const double e = v - 10.0;
        velt = p.w_vel * (e * e);                             // :236
        acct = 0.0;
        if (i > 0) {                                          // (added up only where dtd[i] > 0, like the reference's test)
            const double vp = v0 + dv * p.alpha[i - 1];
            const double a = (v - vp) / p.dtd[i];
            acct = p.w_acc * (a * a);                         // :244
        }
    };
    // the cost chains of one pair from its terms: S_v over all waypoints, then the acceleration terms on top (:235-244)
    auto cost_chains = [&](double velt, double acct, unsigned long long hasm, double* b) {
        double sv = 0.0;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (i < n) sv = sv + bcast(velt, i);
        double run = sv, sa = 0.0;
#pragma unroll
        for (int i = 1; i < 64; ++i)
            if (i < n && ((hasm >> i) & 1ull)) {
                const double term = bcast(acct, i);
                run = run + term, sa = sa + term;
            }
        if (lane == 0) b[0] = sv, b[1] = sa, b[2] = run;
    };

    // ---- phase 1 ---------------------------------------------------------------------------------
    const bool fast = n <= 64 && P <= NW;              // terms in registers, one pair per wave
    double k_velt = 0.0, k_acct = 0.0;                 // this wave's pair: cost terms kept for after the barrier
    unsigned long long k_has = 0;
    bool k_live = false;
    if (fast) {
        if (wid < P) {
            const int g = wid / 3, k = wid - g * 3, f = f0 + g;
            if (f < n_states) {
                const double v0 = state[(size_t)f * 4 + 3];
                const double vt = 8.0 + 2.0 * (double)k;      // [8.0, 10.0, 12.0]  (:280)
                const double dv = vt - v0;
                double* o = vs + ((size_t)(g * 3 + k) * n) * 2;
                double v = 0.0, vdt = 0.0;
                bool has = false;
                if (lane < n) terms(v0, dv, lane, v, vdt, k_velt, k_acct), has = lane > 0 && p.dtd[lane] > 0.0;
                k_has = __ballot(has), k_live = true;
                double sacc = 0.0, my_s = 0.0;
#pragma unroll
                for (int i = 1; i < 64; ++i)
                    if (i < n) {
                        sacc = sacc + bcast(vdt, i);          // :157
                        my_s = lane == i ? sacc : my_s;
                    }
                if (lane < n) o[2 * lane] = v, o[2 * lane + 1] = my_s;
            }
        }
    } else {
        for (int pr = wid; pr < P; pr += NW) {
            const int g = pr / 3, k = pr - g * 3, f = f0 + g;
            if (f >= n_states) continue;
            const double v0 = state[(size_t)f * 4 + 3];
            const double vt = 8.0 + 2.0 * (double)k;
            const double dv = vt - v0;
            double* o = vs + ((size_t)(g * 3 + k) * n) * 2;
            double* b = base + (g * 3 + k) * 3;
            double* tv = stage;                               // [n] v_i dt | [n] velocity cost term | [n] acceleration cost term
            for (int i = lane; i < n; i += 64) {
                double v, vdt, velt, acct;
                terms(v0, dv, i, v, vdt, velt, acct);
                o[2 * i] = v, tv[i] = vdt, tv[n + i] = velt, tv[2 * n + i] = acct;
            }
            wave_lds_fence();
            if (lane == 0) {
                double sacc = 0.0, sv = 0.0;
                o[1] = 0.0;
                sv = sv + tv[n];
                for (int i = 1; i < n; ++i) {
                    sacc = sacc + tv[i];                      // :157
                    o[2 * i + 1] = sacc;
                    sv = sv + tv[n + i];
                }
                double run = sv, sa = 0.0;
                for (int i = 1; i < n; ++i) {
                    if (p.dtd[i] > 0.0) {
                        const double term = tv[2 * n + i];
                        run = run + term, sa = sa + term;
                    }
                }
                b[0] = sv, b[1] = sa, b[2] = run;
            }
            wave_lds_fence();
        }
    }
    // heading terms: the last wave (free in phase 1 when the pairs are fewer than the waves), two lanes per state -- lane 2 g takes
    // sin / cos of the heading, lane 2 g + 1 of heading + pi/2 (one sincos each, side by side)
    if (wid == NW - 1 && lane < 2 * G) {
        const int g = lane >> 1, f = f0 + g;
        if (f < n_states) {
            const double h0 = state[(size_t)f * 4 + 2];
            double* tg = trig + g * 8;
            double sn, cs;
            if (lane & 1) {
                sincos(h0 + 1.5707963267948966, &sn, &cs);       // heading0 + np.pi/2  (:179)
                tg[4] = cs, tg[5] = sn;
            } else {
                sincos(h0, &sn, &cs);
                tg[0] = state[(size_t)f * 4 + 0], tg[1] = state[(size_t)f * 4 + 1];
                tg[2] = cs, tg[3] = sn, tg[6] = h0;
            }
        }
    }
    __syncthreads();
    if (k_live) cost_chains(k_velt, k_acct, k_has, base + wid * 3);        // consumed in phase 3

    // ---- phase 2 ---------------------------------------------------------------------------------
    // the per-waypoint constants of this lane's first waypoint, loaded once (not once per trajectory behind the LDS fences)
    const double q_l = lane < n ? p.q[lane] : 0.0, q_l1 = lane + 1 < n ? p.q[lane + 1] : 0.0, t_l = lane < n ? p.t[lane] : 0.0;
    const bool extra = n_ref > 0 || n_obs > 0;
    // trajectories are dealt round-robin starting BEHIND the waves that still have cost chains to add up
    for (int j = fast ? (wid - P % NW + NW) % NW : wid; j < G * C; j += NW) {
        const int g = j / C, c = j - g * C, f = f0 + g;
        if (f >= n_states) continue;
        const int li = c / 3, k = c - li * 3;
        const double df = p.lat[li];
        const double* tg = trig + g * 8;
        const double x0 = tg[0], y0 = tg[1], cs = tg[2], sn = tg[3], c2 = tg[4], s2 = tg[5], h0 = tg[6];
        const double* o = vs + ((size_t)(g * 3 + k) * n) * 2;

        for (int i = lane; i < n; i += 64) {
            const double v = o[2 * i], s = o[2 * i + 1];
            const bool first = i == lane;
            const double d = df * (first ? q_l : p.q[i]);
            double x = x0 + s * cs, y = y0 + s * sn;            // :175-176
            x = x + d * c2, y = y + d * s2;                     // :179-180
            double hd = 0.0;
            if (i < n - 1) {
                const double s1 = o[2 * i + 3], d1 = df * (first ? q_l1 : p.q[i + 1]);
                double x1 = x0 + s1 * cs, y1 = y0 + s1 * sn;
                x1 = x1 + d1 * c2, y1 = y1 + d1 * s2;
                hd = atan2_fast(y1 - y, x1 - x, p.atq);         // :188
            }
            double* w = stage + (size_t)i * 6;
            w[0] = x, w[1] = y, w[2] = hd, w[3] = v, w[4] = first ? t_l : p.t[i], w[5] = 0.0;
        }
        wave_lds_fence();
        double lat_sum = 0.0, curv_sum = 0.0, obs_sum = 0.0;
        for (int i = lane; i < n; i += 64) {
            double* w = stage + (size_t)i * 6;
            double curv = 0.0;
            if (i > 0 && i < n - 1) {
                const double hp = stage[(size_t)(i - 1) * 6 + 2];
                curv = (w[2] - hp) / (w[3] * p.dt + 1e-6);      // :196
                w[5] = curv;
            }
            curv_sum += p.w_curv * (curv * curv);               // :248
            const double x = w[0], y = w[1];
            if (n_ref > 0) {                                    // :229-231
                double md = INFINITY;
                for (int r = 0; r < n_ref; ++r) {
                    const double dx = ref[2 * r] - x, dy = ref[2 * r + 1] - y;
                    const double dd = sqrt(dx * dx + dy * dy);
                    md = dd < md ? dd : md;
                }
                lat_sum += p.w_lat * (md * md);
            }
            for (int q = 0; q < n_obs; ++q) {                   // :253-259
                const double ox = obs[3 * q], oy = obs[3 * q + 1], rad = obs[3 * q + 2];
                const double ex = x - ox, ey = y - oy;
                const double dist = sqrt(ex * ex + ey * ey);
                if (dist < rad * 2.0) obs_sum += 1000.0 * (rad * 2.0 - dist);
                else if (dist < rad * 4.0) obs_sum += 10.0 / (dist - rad + 0.1);
            }
        }
        wave_lds_fence();
        if (lane == 0) stage[(size_t)(n - 1) * 6 + 2] = n > 1 ? stage[(size_t)(n - 2) * 6 + 2] : h0;   // :190
        curv_sum = wave_sum_dpp(curv_sum);
        if (extra) lat_sum = wave_sum_dpp(lat_sum), obs_sum = wave_sum_dpp(obs_sum);
        if (lane == 0) costs[g * C + c] = lat_sum, costs[CS + g * C + c] = curv_sum, costs[2 * CS + g * C + c] = obs_sum;
        wave_lds_fence();
        if (wp) {
            const double2* src = reinterpret_cast<const double2*>(stage);
            double2* dst = reinterpret_cast<double2*>(wp + ((size_t)f * C + c) * n * 6);
            for (int q = lane; q < n * 3; q += 64) dst[q] = src[q];
        }
        wave_lds_fence();
    }
    __syncthreads();

    // ---- phase 3: costs, stable ascending rank (:300) -------------------------------------------------------------------------
    for (int idx = tid; idx < G * C; idx += NW * 64) {
        const int g = idx / C, c = idx - g * C, k = c % 3;
        const double* b = base + (g * 3 + k) * 3;
        // reference accumulates [ref-path] -> velocity -> acceleration -> curvature -> obstacles
        const double va = n_ref > 0 ? (costs[idx] + b[0]) + b[1] : b[2];
        const double total = (va + costs[CS + idx]) + costs[2 * CS + idx];
        costs[idx] = total;
    }
    __syncthreads();
    for (int idx = tid; idx < G * C; idx += NW * 64) {
        const int g = idx / C, c = idx - g * C, f = f0 + g;
        if (f >= n_states) continue;
        const double* cc = costs + g * C;
        const double mine = cc[c];
        int rank = 0;
        for (int o2 = 0; o2 < C; ++o2) {
            const double v = cc[o2];
            rank += (v < mine || (v == mine && o2 < c)) ? 1 : 0;
        }
        cost[(size_t)f * C + c] = mine;
        order[(size_t)f * C + rank] = c;
    }
}

template <int G, int NW>
__global__ void __launch_bounds__(NW * 64) planner_kernel(PlanParams p, int n_states, const double* __restrict__ state,
                                                          const double* __restrict__ ref, int n_ref,
                                                          const double* __restrict__ obs, int n_obs,
                                                          double* __restrict__ wp, double* __restrict__ cost,
                                                          int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    plan_block<G, NW>(p, blockIdx.x * G, n_states, state, ref, n_ref, obs, n_obs, wp, cost, order, sm);
}

// Output ring of planner_wave_kernel: 256 units of 16 B.  It holds at most 63 carried units + one tile of
// 3n <= 192 units; FPW*3*n <= 384 doubles of phase-1 scratch also fit.
constexpr int RING_UNITS = 256, RING_DOUBLES = RING_UNITS * 2;

// Throughput variant for large batches and n <= 64: every wave is autonomous (no workgroup barrier).
// A wave owns FPW consecutive start states; lane = waypoint index, so the per-waypoint constants
// (1-exp(-t_i), quintic blend, timestamp) sit in registers for the whole kernel.
//   1a  all lanes: velocity profile + acceleration cost terms of the 3*FPW (state, speed) pairs
//   1b  3*FPW lanes: the reference's sequential sums (prefix s_i, velocity cost, acceleration cost)
//   2   for each of the FPW*C trajectories: positions, heading, curvature, cost, AoS tile -> HBM
//   3   stable rank of each state's C costs
template <int FPW, bool EXTRA>     // EXTRA: a reference path and/or obstacles take part in the cost
__global__ void __launch_bounds__(256) planner_wave_kernel(PlanParams p, int n_states,
                                                           const double* __restrict__ state,
                                                           const double* __restrict__ ref, int n_ref,
                                                           const double* __restrict__ obs, int n_obs,
                                                           double* __restrict__ wp, double* __restrict__ cost,
                                                           int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = p.n, C = p.C;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t per_wave = (size_t)FPW * 3 * n * 2 + RING_DOUBLES + even_up(FPW * C) + FPW * 3 * 4 + FPW * 8;
    double* vs = sm + wid * per_wave;                  // [FPW*3][n][2]  (v, s)
    double* stage = vs + (size_t)FPW * 3 * n * 2;      // output ring (256 x 16 B); phase 1: acc terms [FPW*3][n]
    double* costs = stage + RING_DOUBLES;              // [FPW][C]
    double* base = costs + even_up(FPW * C);           // [FPW*3][4]  S_v, S_a, running
    double* trig = base + FPW * 3 * 4;                 // [FPW][8]
    const long long gw = (long long)blockIdx.x * 4 + wid;
    const long long f0l = gw * FPW;
    if (f0l >= n_states) return;
    const int f0 = (int)f0l;
    const int nf = (n_states - f0) < FPW ? (n_states - f0) : FPW;    // states this wave really has

    // All global reads of this wave happen here, in one batch: per-lane table entries (lane = waypoint
    // index), the lateral offsets (lane = sample index) and the wave's start states.  Nothing below
    // reads memory again, so the loops never queue behind the store traffic.
    const bool in = lane < n;
    const int li_c = in ? lane : n - 1;                 // clamped waypoint index of this lane
    const int ln_c = (lane + 1 < n) ? lane + 1 : n - 1; // clamped index of the next waypoint
    const int lp_c = li_c > 0 ? li_c - 1 : 0;
    const double q_i = p.q[li_c], q_n = p.q[ln_c], t_i = p.t[li_c];
    const double al_i = p.alpha[li_c], al_p = p.alpha[lp_c], dtd_i = p.dtd[li_c];
    const double lat_l = p.lat[lane < p.n_lat ? lane : 0];
    double st_x[FPW], st_y[FPW], st_h[FPW], st_v[FPW];
#pragma unroll
    for (int g = 0; g < FPW; ++g) {
        const double4 sv4 = reinterpret_cast<const double4*>(state)[f0 + (g < nf ? g : 0)];
        st_x[g] = sv4.x, st_y[g] = sv4.y, st_h[g] = sv4.z, st_v[g] = sv4.w;
    }

    // ---- 1a: lane = waypoint; velocity profile and acceleration cost term of every (state, speed) -----
#pragma unroll
    for (int g = 0; g < FPW; ++g) {
        if (g < nf) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int pair = g * 3 + k;
                const double v0 = st_v[g];
                const double dv = (8.0 + 2.0 * (double)k) - v0;
                const double v = v0 + dv * al_i;                       // :153-154
                const double vp = v0 + dv * al_p;
                const double a = (v - vp) / dtd_i;
                const double acc = (lane > 0 && dtd_i > 0.0) ? p.w_acc * (a * a) : 0.0;   // :240-244
                if (in) {
                    vs[((size_t)pair * n + lane) * 2] = v;
                    stage[(size_t)pair * n + lane] = acc;
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < FPW; ++g) {
        if (g < nf && lane == 32 + g) {
            const double h0 = st_h[g];
            const double hp = h0 + 1.5707963267948966;
            double* tg = trig + g * 8;
            tg[0] = st_x[g], tg[1] = st_y[g];
            tg[2] = cos(h0), tg[3] = sin(h0), tg[4] = cos(hp), tg[5] = sin(hp), tg[6] = h0;
        }
    }
    wave_lds_fence();
    // ---- 1b: 3*nf lanes run the reference's sequential sums over LDS-resident terms -----------------
    if (lane < nf * 3) {
        double* o = vs + (size_t)lane * n * 2;
        const double* ac = stage + (size_t)lane * n;
        double s = 0.0, sv = 0.0;
        for (int i = 0; i < n; ++i) {
            const double v = o[2 * i];
            if (i > 0) s = s + v * p.dt;                               // :157
            o[2 * i + 1] = s;
            const double e = v - 10.0;
            sv = sv + p.w_vel * (e * e);                               // :236
        }
        double run = sv, sa = 0.0;
        for (int i = 1; i < n; ++i) {                                  // zero terms stand for skipped ones
            const double term = ac[i];
            run = run + term, sa = sa + term;
        }
        double* b = base + lane * 4;
        b[0] = sv, b[1] = sa, b[2] = run;
    }
    wave_lds_fence();

    // ---- 2 -----------------------------------------------------------------------------------------
    // Lanes >= n run the same arithmetic on a clamped index (no predication inside the loop); their
    // results are dropped at the sum and at the store.
    // Output stream.  The wave's trajectories are contiguous in HBM (memory order), 3n 16-byte units each.
    // Tiles are appended to an LDS ring in stream order and leave as FULL 1-KB wave stores aligned in
    // absolute address (only the very first and last store of the wave are partial): a 2448-B tile written
    // as 64+64+25 lanes straddles cache lines at both ends and reaches 5.3 TB/s store-only, aligned full
    // stores 6.0 (tools/wpattern.hip, tools/wfill.hip).  Software pipeline: the chunks completed by tile
    // c-1 are read from the ring at the top of iteration c and stored after its arithmetic.
    const int n3 = n * 3;                                   // units per trajectory (153 for n = 51)
    const size_t G0 = (size_t)(reinterpret_cast<uintptr_t>(wp) >> 4) + (size_t)f0 * C * n3;   // unit address of the stream
    const int A = (int)(G0 & 63);                           // stream unit u sits at virtual unit v = u + A
    double2* const gch = reinterpret_cast<double2*>((G0 - (size_t)A) << 4);                   // virtual unit 0
    double2* const ring = reinterpret_cast<double2*>(stage);
    int ti = 0, jdone = 0;                                  // tiles staged, chunks stored
    double2 r0 = make_double2(0.0, 0.0), r1 = r0, r2 = r0;
    for (int g = 0; g < nf; ++g) {
        const int f = f0 + g;
        const double* tg = trig + g * 8;
        const double x0 = tg[0], y0 = tg[1], cs = tg[2], sn = tg[3], c2 = tg[4], s2 = tg[5], h0 = tg[6];
        // per speed k: everything the 7 lateral samples share, kept in registers so that the trajectories can
        // be produced in memory order c = li*3 + k (a wave then writes its 51 KB sequentially; the speed-outer
        // order hops 7344 B between tiles and costs a quarter of the write rate -- tools/wpattern.hip)
        double kv[3], kbx[3], kby[3], kbx1[3], kby1[3], kden[3], krden[3], kb0[3], kb1[3], kb2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double* o = vs + ((size_t)(g * 3 + k) * n) * 2;
            const double v = o[2 * li_c], s = o[2 * li_c + 1], s1 = o[2 * ln_c + 1];
            kv[k] = v;
            kbx[k] = x0 + s * cs, kby[k] = y0 + s * sn;              // :175-176
            kbx1[k] = x0 + s1 * cs, kby1[k] = y0 + s1 * sn;
            kden[k] = v * p.dt + 1e-6;                               // :196 denominator
            krden[k] = 1.0 / kden[k];                                // shared by the n_lat trajectories of this speed
            const double* b = base + (g * 3 + k) * 4;
            kb0[k] = b[0], kb1[k] = b[1], kb2[k] = b[2];
        }
        for (int li = 0; li < p.n_lat; ++li) {
            const double df = readlane_f64(lat_l, li);
            const double d = df * q_i, d1 = df * q_n;
            const double dxc = d * c2, dyc = d * s2, dxc1 = d1 * c2, dyc1 = d1 * s2;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = li * 3 + k;
                const double v = kv[k], bx = kbx[k], by = kby[k], bx1 = kbx1[k], by1 = kby1[k];
                const double den = kden[k], rden = krden[k], b0 = kb0[k], b1 = kb1[k], b2 = kb2[k];
                const int jend = (A + ti * n3) >> 6;       // chunks complete once tiles < ti are staged
                const int np = wp ? jend - jdone : 0;        // 0 (first tile) .. 3
                if (np > 0) {            // issue the LDS reads early; consumed after the math
                    wave_lds_fence();
                    const int u0 = jdone * 64 + lane;
                    r0 = ring[u0 & (RING_UNITS - 1)];
                    if (np > 1) r1 = ring[(u0 + 64) & (RING_UNITS - 1)];
                    if (np > 2) r2 = ring[(u0 + 128) & (RING_UNITS - 1)];
                    wave_lds_fence();
                }
                const double x = bx + dxc, y = by + dyc;             // :179-180
                const double x1 = bx1 + dxc1, y1 = by1 + dyc1;
                double hd = atan2_fast(y1 - y, x1 - x, p.atq);       // :188
                const double hprev = dpp_mov_f64<0x138>(hd);         // lane i <- lane i-1
                // (hd - hprev) / den, correctly rounded from the correctly rounded reciprocal (Markstein):
                // q0 = a*r, q = q0 + (a - q0*den)*r
                const double dh = hd - hprev;
                const double cq = dh * rden;
                double curv = __builtin_fma(__builtin_fma(-cq, den, dh), rden, cq);
                curv = (lane > 0 && lane < n - 1) ? curv : 0.0;
                if (lane == n - 1) hd = n > 1 ? hprev : h0;          // :190
                double lat_sum = 0.0, obs_sum = 0.0;
                if (EXTRA && n_ref > 0) {
                    double md = INFINITY;
                    for (int r = 0; r < n_ref; ++r) {
                        const double dx = ref[2 * r] - x, dy = ref[2 * r + 1] - y;
                        const double dd = sqrt(dx * dx + dy * dy);
                        md = dd < md ? dd : md;
                    }
                    lat_sum = wave_sum_dpp(in ? p.w_lat * (md * md) : 0.0);
                }
                if (EXTRA && n_obs > 0) {
                    for (int q = 0; q < n_obs; ++q) {
                        const double ox = obs[3 * q], oy = obs[3 * q + 1], rad = obs[3 * q + 2];
                        const double ex = x - ox, ey = y - oy;
                        const double dist = sqrt(ex * ex + ey * ey);
                        if (dist < rad * 2.0) obs_sum += 1000.0 * (rad * 2.0 - dist);
                        else if (dist < rad * 4.0) obs_sum += 10.0 / (dist - rad + 0.1);
                    }
                    obs_sum = wave_sum_dpp(in ? obs_sum : 0.0);
                }
                const double curv_sum = wave_sum_dpp(in ? p.w_curv * (curv * curv) : 0.0);
                if (lane == 0) {
                    const double va = (EXTRA && n_ref > 0) ? (lat_sum + b0) + b1 : b2;
                    costs[g * C + c] = EXTRA ? (va + curv_sum) + obs_sum : va + curv_sum;
                }
                if (wp) {
                    // stream out the chunks read at the top of this iteration, then append this tile
                    if (np > 0) {
                        const int v0 = jdone * 64 + lane;
                        if (v0 >= A) gch[v0] = r0;                       // only chunk 0 has units before the stream
                        if (np > 1) gch[v0 + 64] = r1;
                        if (np > 2) gch[v0 + 128] = r2;
                        jdone = jend;
                    }
                    if (in) {
                        const int u = A + ti * n3 + lane * 3;
                        ring[u & (RING_UNITS - 1)] = make_double2(x, y);
                        ring[(u + 1) & (RING_UNITS - 1)] = make_double2(hd, v);
                        ring[(u + 2) & (RING_UNITS - 1)] = make_double2(t_i, curv);
                    }
                    ++ti;
                }
            }
        }
    }
    if (wp) {                 // drain: remaining full chunks and the partial tail
        wave_lds_fence();
        const int vend = A + ti * n3;
        for (int j = jdone; j * 64 < vend; ++j) {
            const int v = j * 64 + lane;
            if (v >= A && v < vend) gch[v] = ring[v & (RING_UNITS - 1)];
        }
    }
    wave_lds_fence();
    // ---- 3 -----------------------------------------------------------------------------------------
    for (int idx = lane; idx < nf * C; idx += 64) {
        const int g = idx / C, c = idx - g * C, f = f0 + g;
        const double* cc = costs + g * C;
        const double mine = cc[c];
        int rank = 0;
        for (int o2 = 0; o2 < C; ++o2) {
            const double vv = cc[o2];
            rank += (vv < mine || (vv == mine && o2 < c)) ? 1 : 0;
        }
        cost[(size_t)f * C + c] = mine;
        order[(size_t)f * C + rank] = c;
    }
}

// generate_polynomial_trajectory for one arbitrary (df, vt): one wave per trajectory.
__global__ void __launch_bounds__(64) planner_generate_kernel(PlanParams p, int n_traj, const double* __restrict__ state,
                                                              const double* __restrict__ dfs,
                                                              const double* __restrict__ vts, double* __restrict__ wp) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = p.n, lane = threadIdx.x, j = blockIdx.x;
    if (j >= n_traj) return;
    double* vs = sm;                 // [n][2]
    double* stage = sm + 2 * n;      // [n][6]
    const double x0 = state[4 * j], y0 = state[4 * j + 1], h0 = state[4 * j + 2], v0 = state[4 * j + 3];
    const double df = dfs[j], vt = vts[j];
    if (lane == 0) {
        const double dv = vt - v0;
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            const double v = v0 + dv * p.alpha[i];
            if (i > 0) s = s + v * p.dt;
            vs[2 * i] = v, vs[2 * i + 1] = s;
        }
    }
    wave_lds_fence();
    const double hp2 = h0 + 1.5707963267948966;
    const double cs = cos(h0), sn = sin(h0), c2 = cos(hp2), s2 = sin(hp2);
    for (int i = lane; i < n; i += 64) {
        const double v = vs[2 * i], s = vs[2 * i + 1];
        const double d = df * p.q[i];
        double x = x0 + s * cs, y = y0 + s * sn;
        x = x + d * c2, y = y + d * s2;
        double hd = 0.0;
        if (i < n - 1) {
            const double s1 = vs[2 * i + 3], d1 = df * p.q[i + 1];
            double x1 = x0 + s1 * cs, y1 = y0 + s1 * sn;
            x1 = x1 + d1 * c2, y1 = y1 + d1 * s2;
            hd = atan2(y1 - y, x1 - x);
        }
        double* w = stage + (size_t)i * 6;
        w[0] = x, w[1] = y, w[2] = hd, w[3] = v, w[4] = p.t[i], w[5] = 0.0;
    }
    wave_lds_fence();
    for (int i = lane; i < n; i += 64)
        if (i > 0 && i < n - 1) {
            double* w = stage + (size_t)i * 6;
            w[5] = (w[2] - stage[(size_t)(i - 1) * 6 + 2]) / (w[3] * p.dt + 1e-6);
        }
    wave_lds_fence();
    if (lane == 0) stage[(size_t)(n - 1) * 6 + 2] = n > 1 ? stage[(size_t)(n - 2) * 6 + 2] : h0;
    wave_lds_fence();
    double* dst = wp + (size_t)j * n * 6;
    for (int q = lane; q < n * 6; q += 64) dst[q] = stage[q];
}

// evaluate_trajectory_cost for an arbitrary trajectory, strictly in the reference's accumulation order
// (one thread per trajectory; this is a utility entry, the hot path is planner_kernel).
__global__ void planner_evaluate_kernel(PlanParams p, int n_traj, int n_wp, const double* __restrict__ wp,
                                        const double* __restrict__ ref, int n_ref, const double* __restrict__ obs,
                                        int n_obs, double* __restrict__ cost) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_traj) return;
    if (n_wp == 0) {
        cost[j] = INFINITY;
        return;
    }
    const double* w = wp + (size_t)j * n_wp * 6;
    double c = 0.0;
    if (n_ref > 0)
        for (int i = 0; i < n_wp; ++i) {
            double md = INFINITY;
            for (int r = 0; r < n_ref; ++r) {
                const double dx = ref[2 * r] - w[6 * i], dy = ref[2 * r + 1] - w[6 * i + 1];
                const double dd = sqrt(dx * dx + dy * dy);
                md = dd < md ? dd : md;
            }
            c = c + p.w_lat * (md * md);
        }
    for (int i = 0; i < n_wp; ++i) {
        const double e = w[6 * i + 3] - 10.0;
        c = c + p.w_vel * (e * e);
    }
    for (int i = 1; i < n_wp; ++i) {
        const double dtt = w[6 * i + 4] - w[6 * (i - 1) + 4];
        if (dtt > 0.0) {
            const double a = (w[6 * i + 3] - w[6 * (i - 1) + 3]) / dtt;
            c = c + p.w_acc * (a * a);
        }
    }
    for (int i = 0; i < n_wp; ++i) c = c + p.w_curv * (w[6 * i + 5] * w[6 * i + 5]);
    for (int q = 0; q < n_obs; ++q) {
        const double ox = obs[3 * q], oy = obs[3 * q + 1], rad = obs[3 * q + 2];
        for (int i = 0; i < n_wp; ++i) {
            const double ex = w[6 * i] - ox, ey = w[6 * i + 1] - oy;
            const double dist = sqrt(ex * ex + ey * ey);
            if (dist < rad * 2.0) c = c + 1000.0 * (rad * 2.0 - dist);
            else if (dist < rad * 4.0) c = c + 10.0 / (dist - rad + 0.1);
        }
    }
    cost[j] = c;
}

}  // namespace

static void fill_params(const av_ctx* ctx, PlanParams& p) {
    const int n = ctx->n_points;
    p.n = n, p.n_lat = ctx->n_lat, p.C = ctx->n_cand;
    p.dt = ctx->pcfg.dt, p.H = ctx->pcfg.planning_horizon;
    p.w_lat = ctx->pcfg.w_lateral, p.w_vel = ctx->pcfg.w_velocity, p.w_acc = ctx->pcfg.w_acceleration;
    p.w_curv = ctx->pcfg.w_curvature;
    p.t = ctx->d_ptab, p.alpha = p.t + n, p.q = p.alpha + n, p.dtd = p.q + n, p.lat = p.dtd + n;
    for (int k = 0; k < 20; ++k) p.atq[k] = ATAN_Q[k];
}

#ifndef AVHOT_DEVICE_ONLY      // (step.hip includes this file for its device code only)


extern "C" {

int av_planner_generate(av_ctx* ctx, av_stream_t stream, int n_traj, const double* state,
                        const double* end_lateral_offset, const double* target_velocity, double* waypoints) {
    AV_REQUIRE(ctx && state && end_lateral_offset && target_velocity && waypoints, AV_EINVAL,
               "av_planner_generate: null argument");
    AV_REQUIRE(ctx->planner_ready, AV_ESTATE, "av_planner_generate: call av_planner_configure first");
    AV_REQUIRE(n_traj > 0, AV_EINVAL, "av_planner_generate: n_traj must be > 0");
    PlanParams p;
    fill_params(ctx, p);
    hipLaunchKernelGGL(planner_generate_kernel, dim3(n_traj), dim3(64), (size_t)p.n * 8 * 8, as_stream(stream), p,
                       n_traj, state, end_lateral_offset, target_velocity, waypoints);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_planner_evaluate(av_ctx* ctx, av_stream_t stream, int n_traj, int n_wp, const double* waypoints,
                        const double* ref_path, int n_ref, const double* obstacles, int n_obs, double* cost) {
    AV_REQUIRE(ctx && cost && (waypoints || n_wp == 0), AV_EINVAL, "av_planner_evaluate: null argument");
    AV_REQUIRE(ctx->planner_ready, AV_ESTATE, "av_planner_evaluate: call av_planner_configure first");
    AV_REQUIRE(n_traj > 0 && n_wp >= 0, AV_EINVAL, "av_planner_evaluate: bad sizes");
    AV_REQUIRE(n_ref >= 0 && n_obs >= 0 && (n_ref == 0 || ref_path) && (n_obs == 0 || obstacles), AV_EINVAL,
               "av_planner_evaluate: ref_path/obstacles pointer missing");
    PlanParams p;
    fill_params(ctx, p);
    hipLaunchKernelGGL(planner_evaluate_kernel, dim3((n_traj + 63) / 64), dim3(64), 0, as_stream(stream), p, n_traj,
                       n_wp, waypoints, ref_path, n_ref, obstacles, n_obs, cost);
    AV_LAUNCH_CHECK();
    return AV_OK;
}

int av_planner_configure(av_ctx* ctx, const av_planner_cfg* cfg) {
    AV_REQUIRE(ctx && cfg, AV_EINVAL, "av_planner_configure: null argument");
    AV_REQUIRE(cfg->dt > 0.0 && cfg->planning_horizon > 0.0, AV_EINVAL, "av_planner_configure: dt and horizon must be > 0");
    AV_REQUIRE(cfg->num_samples >= 1 && cfg->num_samples <= 64, AV_EINVAL,
               "av_planner_configure: num_samples %d not in [1,64]", cfg->num_samples);
    const double H = cfg->planning_horizon;
    const int n = (int)(H / cfg->dt) + 1;                               // :143
    AV_REQUIRE(n >= 1 && n <= 256, AV_EINVAL, "av_planner_configure: %d waypoints per trajectory not in [1,256]", n);
    const int nl = cfg->num_samples;
    std::vector<double> tab((size_t)4 * n + nl);
    double *t = tab.data(), *alpha = t + n, *q = alpha + n, *dtd = q + n, *lat = dtd + n;
    // numpy.linspace(0, H, n): arange * step + start, last element forced to stop      (:144)
    if (n == 1) {
        t[0] = 0.0;
    } else {
        const double step = (H - 0.0) / (double)(n - 1);
        for (int i = 0; i < n; ++i) t[i] = (step == 0.0 ? ((double)i / (double)(n - 1)) * H : (double)i * step) + 0.0;
        t[n - 1] = H;
    }
    for (int i = 0; i < n; ++i) {
        alpha[i] = 1.0 - std::exp(-t[i]);                                // :153
        double tau = t[i] / H;                                           // :166-167
        tau = tau < 0.0 ? 0.0 : (tau > 1.0 ? 1.0 : tau);
        q[i] = 10.0 * std::pow(tau, 3.0) - 15.0 * std::pow(tau, 4.0) + 6.0 * std::pow(tau, 5.0);   // :169
        dtd[i] = i > 0 ? t[i] - t[i - 1] : 0.0;
    }
    if (nl == 1) {
        lat[0] = -3.5;                                                   // linspace(-3.5, 3.5, 1)
    } else {
        const double step = (3.5 - (-3.5)) / (double)(nl - 1);           // :279
        for (int i = 0; i < nl; ++i) lat[i] = (double)i * step + (-3.5);
        lat[nl - 1] = 3.5;
    }
    AV_HIP(hipSetDevice(ctx->device));
    if (ctx->d_ptab) {
        AV_HIP(hipFree(ctx->d_ptab));
        ctx->d_ptab = nullptr;
    }
    AV_HIP(hipMalloc(&ctx->d_ptab, tab.size() * sizeof(double)));
    AV_HIP(hipMemcpy(ctx->d_ptab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->pcfg = *cfg;
    ctx->n_points = n, ctx->n_lat = nl, ctx->n_cand = 3 * nl;
    ctx->planner_ready = true;
    return AV_OK;
}

int av_planner_dims(const av_ctx* ctx, int* n_points, int* n_candidates) {
    AV_REQUIRE(ctx && n_points && n_candidates, AV_EINVAL, "av_planner_dims: null argument");
    AV_REQUIRE(ctx->planner_ready, AV_ESTATE, "av_planner_dims: call av_planner_configure first");
    *n_points = ctx->n_points, *n_candidates = ctx->n_cand;
    return AV_OK;
}

int av_planner_plan(av_ctx* ctx, av_stream_t stream, int n_states, const double* state, const double* ref_path,
                    int n_ref, const double* obstacles, int n_obs, double* waypoints, double* cost, int32_t* order) {
    AV_REQUIRE(ctx && state && cost && order, AV_EINVAL, "av_planner_plan: null argument");
    AV_REQUIRE(ctx->planner_ready, AV_ESTATE, "av_planner_plan: call av_planner_configure first");
    AV_REQUIRE(n_states > 0, AV_EINVAL, "av_planner_plan: n_states must be > 0");
    AV_REQUIRE(n_ref >= 0 && n_obs >= 0 && (n_ref == 0 || ref_path) && (n_obs == 0 || obstacles), AV_EINVAL,
               "av_planner_plan: ref_path/obstacles pointer missing");
    AV_REQUIRE(n_ref != 1, AV_EINVAL, "av_planner_plan: a reference path needs >= 2 points (set_reference_path ignores shorter)");
    const int n = ctx->n_points, C = ctx->n_cand;
    PlanParams p;
    fill_params(ctx, p);
    hipStream_t st = as_stream(stream);
    if (n <= 64 && n_states >= 1024) {
        // two states per wave (103-KB output regions); one per wave -- 51-KB regions, twice the waves -- measured the same
        // (59.4-60.7 % of HBM peak at 16 384 states either way, alternating runs on one box)
        constexpr int FPW = 2;
        const size_t per_wave = (size_t)FPW * 3 * n * 2 + RING_DOUBLES + even_up(FPW * C) + FPW * 3 * 4 + FPW * 8;
        const size_t lds_w = per_wave * 4 * sizeof(double);
        if (lds_w <= 64 * 1024) {
            const int grid_w = (n_states + 4 * FPW - 1) / (4 * FPW);
            if (n_ref > 0 || n_obs > 0)
                hipLaunchKernelGGL((planner_wave_kernel<FPW, true>), dim3(grid_w), dim3(256), lds_w, st, p, n_states, state,
                                   ref_path, n_ref, obstacles, n_obs, waypoints, cost, order);
            else
                hipLaunchKernelGGL((planner_wave_kernel<FPW, false>), dim3(grid_w), dim3(256), lds_w, st, p, n_states, state,
                                   ref_path, n_ref, obstacles, n_obs, waypoints, cost, order);
            AV_LAUNCH_CHECK();
            return AV_OK;
        }
    }
    int G = n_states >= 4096 ? 8 : (n_states >= 1024 ? 4 : (n_states >= 512 ? 2 : 1));
    const int NW = G == 1 ? 8 : 4;                      // one state per workgroup: its 3 C trajectories over eight waves
    while (G > 1 && plan_lds_doubles(G, n, C, NW) * 8 > 48 * 1024) G >>= 1;
    const size_t lds = plan_lds_doubles(G, n, C, G == 1 ? 8 : 4) * 8;
    AV_REQUIRE(lds <= 64 * 1024, AV_EINVAL, "av_planner_plan: configuration needs %zu B of LDS", lds);
    const int grid = (n_states + G - 1) / G;
#define AV_PLAN_LAUNCH(GG, NWV)                                                                                        \
    hipLaunchKernelGGL((planner_kernel<GG, NWV>), dim3(grid), dim3(NWV * 64), lds, st, p, n_states, state, ref_path, n_ref,  \
                       obstacles, n_obs, waypoints, cost, order)
    switch (G) {
        case 8: AV_PLAN_LAUNCH(8, 4); break;
        case 4: AV_PLAN_LAUNCH(4, 4); break;
        case 2: AV_PLAN_LAUNCH(2, 4); break;
        default: AV_PLAN_LAUNCH(1, 8); break;
    }
#undef AV_PLAN_LAUNCH
    AV_LAUNCH_CHECK();
    return AV_OK;
}

}  // extern "C"

#endif  // AVHOT_DEVICE_ONLY
