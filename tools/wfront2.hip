// Probe 2: do bursty stores overlap with f64 compute at 16 waves/CU?  (tuning aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(double2* out, int nchunk, int chunk16, int spin, int do_store) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3;
    double2* base = out + gw * (size_t)nchunk * chunk16;
    for (int c = 0; c < nchunk; ++c) {
        for (int s = 0; s < spin; ++s) {     // 4 independent f64 chains
            a0 = a0 * 1.0000001 + 0.5; a1 = a1 * 0.9999999 + 0.25; a2 = a2 * 1.0000002 - 0.5; a3 = a3 * 0.9999998 + 0.125;
        }
        if (do_store)
            for (int q = lane; q < chunk16; q += 64) base[(size_t)c * chunk16 + q] = make_double2(a0 + a1, a2 + a3);
    }
    if (!do_store && a0 + a1 + a2 + a3 == 123.456) base[lane] = make_double2(a0, a1);
    if (threadIdx.x == 0) lds[0] = a0;
}
int main() {
    const int waves = 16384 * 4, nchunk = 42, chunk = 2448;
    const size_t total = (size_t)waves * nchunk * chunk;
    double2* d; hipMalloc(&d, total);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("total %.2f GB\n", total / 1e9);
    for (int spin : {0, 20, 40, 60, 80, 120})
    for (int st = 0; st < 2; ++st) {
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(a);
            k<<<waves / 4, 256, 40 * 1024>>>(d, nchunk, chunk / 16, spin, st);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("spin=%3d store=%d: %.3f ms  %s\n", spin, st, best, st ? "" : "(compute only)");
        fflush(stdout);
    }
    return 0;
}
