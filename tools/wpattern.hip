// Store-pattern microbenchmark for the planner's output stream (not part of the library).
// Every wave writes `tiles` tiles of 2448 B (64+64+25 lanes x 16 B, like one 51-waypoint trajectory) into
// its own region of tiles*2448 B; the tile order inside the region is either sequential or strided by 3
// (the speed-outer order of planner_wave_kernel).  `work` dummy FMAs per tile emulate the compute.
//   hipcc --offload-arch=gfx950 -O3 tools/wpattern.hip -o tools/wpattern && tools/wpattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ORDER>   // 0 sequential, 1 stride-3 (k outer), 2 waves of a block interleave tiles (tile t of wave w at (t*4+w))
__global__ void __launch_bounds__(256) wpat(double2* out, int tiles, int work, long long total_waves) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long gw = (long long)blockIdx.x * 4 + wid;
    if (gw >= total_waves) return;
    const size_t tile16 = 153;     // 16-byte chunks per tile
    double acc = lane * 1e-3;
    for (int it = 0; it < tiles; ++it) {
        int t = it;
        if (ORDER == 1) { const int per = (tiles + 2) / 3; t = (it % per) * 3 + it / per; if (t >= tiles) t = it; }
        size_t base;
        if (ORDER == 2) base = ((size_t)blockIdx.x * 4 * tiles + (size_t)it * 4 + wid) * tile16;
        else base = ((size_t)gw * tiles + t) * tile16;
        for (int k = 0; k < work; ++k) acc = __builtin_fma(acc, 1.0000001, 1e-9);
        const double2 v = make_double2(acc, acc);
        if (ORDER == 3) continue;
        out[base + lane] = v;
        out[base + 64 + lane] = v;
        if (lane < 25) out[base + 128 + lane] = v;
    }
    if (ORDER == 3) {        // the same region as 1024-B aligned chunks (partial head and tail)
        const size_t start = (size_t)gw * tiles * tile16, end = start + (size_t)tiles * tile16;   // in 16-B units
        const double2 v = make_double2(acc, acc);
        for (size_t c = start & ~size_t(63); c < end; c += 64) {
            const size_t a = c + lane;
            if (a >= start && a < end) out[a] = v;
        }
    }
}

int main(int argc, char** argv) {
    const size_t total_tiles = 131072ull * 21;            // one planner launch of 131072 states
    const size_t bytes = total_tiles * 2448;
    double2* buf;
    CK(hipMalloc(&buf, bytes + 4096));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles_opts[] = {1, 3, 21, 42, 168};
    for (int order : {0, 3})
        for (int work : {0})
            for (int tiles : tiles_opts) {
                const long long waves = (long long)(total_tiles / tiles);
                const int grid = (int)((waves + 3) / 4);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(a));
                    if (order == 0) hipLaunchKernelGGL(wpat<0>, dim3(grid), dim3(256), 0, 0, buf, tiles, work, waves);
                    else if (order == 1) hipLaunchKernelGGL(wpat<1>, dim3(grid), dim3(256), 0, 0, buf, tiles, work, waves);
                    else if (order == 3) hipLaunchKernelGGL(wpat<3>, dim3(grid), dim3(256), 0, 0, buf, tiles, work, waves);
                    else hipLaunchKernelGGL(wpat<2>, dim3(grid), dim3(256), 0, 0, buf, tiles, work, waves);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (rep && ms < best) best = ms;
                }
                printf("order %d work %3d tiles/wave %3d (region %7.1f KB): %.3f ms  %.0f GB/s\n", order, work, tiles,
                       tiles * 2448 / 1024.0, best, bytes / (best * 1e-3) / 1e9);
                fflush(stdout);
            }
    return 0;
}
