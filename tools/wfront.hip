// Probe: HBM write rate vs. number of concurrent write fronts and burst size (tuning aid, not product code).
// Each wave owns a contiguous region and writes it in bursts of `chunk` bytes with `spin` dummy work between bursts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(256) wfront(double2* out, size_t region16, int chunk16, int spin, int interleave) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double acc = (double)lane;
    if (!interleave) {
        double2* base = out + gw * region16;
        for (size_t off = 0; off < region16; off += chunk16) {
            for (int s = 0; s < spin; ++s) acc = acc * 1.0000001 + 0.5;
            for (int q = lane; q < chunk16; q += 64) base[off + q] = make_double2(acc, acc);
        }
    } else {
        // the 4 waves of a block write adjacent chunks: block region = 4*region16, chunk index round-robin over waves
        const int wid = threadIdx.x >> 6;
        double2* base = out + (size_t)blockIdx.x * 4 * region16;
        const size_t nchunks = 4 * region16 / chunk16;
        for (size_t c = wid; c < nchunks; c += 4) {
            for (int s = 0; s < spin; ++s) acc = acc * 1.0000001 + 0.5;
            for (int q = lane; q < chunk16; q += 64) base[c * chunk16 + q] = make_double2(acc, acc);
        }
    }
}
int main(int argc, char** argv) {
    const size_t total = (size_t)6 << 30;   // 6 GiB
    double2* d;
    hipMalloc(&d, total);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int chunks[] = {2448, 4896, 9792, 19584, 52224};
    for (int inter = 0; inter < 2; ++inter)
    for (int spin : {0, 400})
    for (int waves : {4096, 16384})
    for (int chunk : chunks) {
        size_t region = total / waves;
        region = region / chunk * chunk;          // multiple of chunk
        const size_t region16 = region / 16; const int chunk16 = chunk / 16;
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(a);
            wfront<<<waves / 4, 256>>>(d, region16, chunk16, spin, inter);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("inter=%d spin=%3d waves=%5d chunk=%6d B: %.3f ms  %.0f GB/s\n", inter, spin, waves, chunk, best,
               (double)region * waves / best / 1e6);
        fflush(stdout);
    }
    return 0;
}
