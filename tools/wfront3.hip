// Probe 3: per-CU store issue rate when the target stays in L2 (no HBM), vs waves per CU.  (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(double2* out, int iters, size_t wrap16, int strided) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double2* base = out + (gw * 153 * 7) % wrap16;
    double a = lane;
    for (int c = 0; c < iters; ++c) {
        a += 1.0;
        if (!strided) {
            base[lane] = make_double2(a, a); base[64 + lane] = make_double2(a, a);
            if (lane < 25) base[128 + lane] = make_double2(a, a);
        } else if (lane < 51) {
            base[lane * 3] = make_double2(a, a); base[lane * 3 + 1] = make_double2(a, a); base[lane * 3 + 2] = make_double2(a, a);
        }
    }
}
int main() {
    double2* d; hipMalloc(&d, (size_t)64 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (int strided = 0; strided < 2; ++strided)
    for (int wgs : {256, 512, 1024, 2048}) {           // 1,2,4,8 WGs (4,8,16,32 waves) per CU
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(a);
            k<<<wgs, 256>>>(d, iters, ((size_t)4 << 20) / 16, strided);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        double bytes = (double)wgs * 4 * iters * 2448;
        printf("strided=%d waves/CU=%2d: %.3f ms  %.0f GB/s  %.1f B/clk/CU @2.1GHz\n", strided, wgs / 64, best,
               bytes / best / 1e6, bytes / best / 1e6 / 256 / 2.1);
    }
    return 0;
}
