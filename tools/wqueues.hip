// Which pairs of HIP streams really run side by side?  K streams are created one after the other and each is used once (that is
// when the runtime gives it a hardware queue).  Then, for every pair (i, j), a spin kernel of 64 workgroups (a quarter of the chip)
// is launched on each and the pair is timed: T if they ran beside each other, 2 T if one waited for the other.
// Prints the matrix; with rocprofv3 --kernel-trace the queue ids can be read next to it (tools/qmap.py).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const int prio_first = argc > 2 ? atoi(argv[2]) : 0;          // 1: stream 0 is created with the highest priority
    std::vector<hipStream_t> st(K);
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int i = 0; i < K; ++i) {
        if (i == 0 && prio_first) (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, greatest);
        else (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 10ull, nullptr);
        (void)hipStreamSynchronize(st[i]);
    }
    const unsigned long long T = 50000ull;                         // 0.5 ms
    printf("pair time in units of one kernel's time (1.0 = side by side, 2.0 = one after the other); %d streams, priorities %d..%d%s\n", K, least,
           greatest, prio_first ? ", stream 0 highest" : "");
    printf("     ");
    for (int j = 0; j < K; ++j) printf("%5d", j);
    printf("\n");
    for (int i = 0; i < K; ++i) {
        printf("%3d: ", i);
        for (int j = 0; j < K; ++j) {
            if (j <= i) { printf("     "); continue; }
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[i], T, nullptr);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[j], T, nullptr);
            (void)hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%5.2f", us / 500.0);
        }
        printf("\n");
    }
    return 0;
}
