// Which property of a plain fill reaches 6.5 TB/s?  Store-only variants over the same 6.8 GB (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A: one 16-B store per thread, block = 256 threads (4 KB per block)
__global__ void __launch_bounds__(256) fillA(double2* out, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) out[i] = make_double2(1.0, 2.0);
}
// B: U stores per thread, block covers U*4 KB contiguous, thread stride 256 (torch-like)
template <int U>
__global__ void __launch_bounds__(256) fillB(double2* out, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; if (i < n16) out[i] = make_double2(1.0, 2.0); }
}
// C: each WAVE owns a contiguous region of R 1-KB chunks and walks it (like the planner's per-wave stream, aligned)
__global__ void __launch_bounds__(256) fillC(double2* out, size_t n16, int R) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int r = 0; r < R; ++r) { const size_t i = (gw * R + r) * 64 + lane; if (i < n16) out[i] = make_double2(1.0, 2.0); }
}
// D: each BLOCK owns R*4 chunks; its 4 waves interleave chunk by chunk (block-linear stream)
__global__ void __launch_bounds__(256) fillD(double2* out, size_t n16, int R) {
    const size_t b = blockIdx.x;
    for (int r = 0; r < R; ++r) { const size_t i = (b * R + r) * 256 + threadIdx.x; if (i < n16) out[i] = make_double2(1.0, 2.0); }
}
// E: persistent grid-stride (grid = 256 CUs * 8 blocks), whole grid sweeps linearly
__global__ void __launch_bounds__(256) fillE(double2* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = make_double2(1.0, 2.0);
}

int main() {
    const size_t bytes = 131072ull * 21 * 2448, n16 = bytes / 16;
    double2* buf; CK(hipMalloc(&buf, bytes + 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9); fflush(stdout);
    };
    run("A  1 store/thread", [&] { hipLaunchKernelGGL(fillA, dim3((n16 + 255) / 256), dim3(256), 0, 0, buf, n16); });
    run("B  4 stores/thread (16 KB/block)", [&] { hipLaunchKernelGGL(fillB<4>, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, buf, n16); });
    run("B  16 stores/thread (64 KB/block)", [&] { hipLaunchKernelGGL(fillB<16>, dim3((n16 + 4095) / 4096), dim3(256), 0, 0, buf, n16); });
    for (int R : {4, 16, 100}) {
        char nm[64]; snprintf(nm, 64, "C  wave walks %d KB", R);
        run(nm, [&] { hipLaunchKernelGGL(fillC, dim3((n16 / 64 / R + 4) / 4), dim3(256), 0, 0, buf, n16, R); });
    }
    for (int R : {4, 16, 100}) {
        char nm[64]; snprintf(nm, 64, "D  block walks %d KB (waves interleaved)", R * 4);
        run(nm, [&] { hipLaunchKernelGGL(fillD, dim3(n16 / 256 / R + 1), dim3(256), 0, 0, buf, n16, R); });
    }
    run("E  persistent grid-stride (2048 blocks)", [&] { hipLaunchKernelGGL(fillE, dim3(2048), dim3(256), 0, 0, buf, n16); });
    run("E  persistent grid-stride (8192 blocks)", [&] { hipLaunchKernelGGL(fillE, dim3(8192), dim3(256), 0, 0, buf, n16); });
    return 0;
}
