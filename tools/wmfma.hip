// Calibration for the YOLO convolution kernels (not part of the library): what one wave per SIMD reaches with
//   A  back-to-back independent v_mfma_f32_16x16x32_f16 out of registers,
//   B  the conv3x3_ws_kernel inner step: MT + NT ds_read_b128 prefetched one step ahead of MT*NT MFMAs,
//   C  the epilogue's arithmetic alone (bias + SiLU on MT*NT*4 values) and D with its 8-byte scattered stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NT>
__global__ void __launch_bounds__(256) mfma_regs(float* out, int iters) {
    half8 A[MT], B[NT];
    for (int i = 0; i < MT; ++i) for (int k = 0; k < 8; ++k) A[i][k] = (_Float16)(threadIdx.x * 0.001f + i);
    for (int i = 0; i < NT; ++i) for (int k = 0; k < 8; ++k) B[i][k] = (_Float16)(threadIdx.x * 0.002f + i);
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.f) out[0] = s;
}

template <int MT, int NT, int PAD>
__global__ void __launch_bounds__(256) mfma_lds(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    constexpr int wrowb = 9 * 64 * 2 + PAD, pixb = 64 * 2 + PAD, PW = 18;
    unsigned char* wts = lsm;
    unsigned char* patch = lsm + 16 * MT * wrowb;
    for (int i = threadIdx.x; i < (16 * MT * wrowb + (4 * NT + 2) * PW * pixb) / 4; i += 256) reinterpret_cast<float*>(lsm)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, h = lane >> 4;
    const unsigned char* arow = wts + l15 * wrowb + 16 * h;
    const unsigned char* brow = patch + (NT * wave * PW + l15) * pixb + 16 * h;
    f32x4 acc[MT][NT];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    auto ld = [&](half8* A, half8* B, int step) {
        const int c0 = (step / 9) * 32, tap = step % 9, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) A[mt] = *reinterpret_cast<const half8*>(arow + mt * 16 * wrowb + (tap * 64 + c0) * 2);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) B[nt] = *reinterpret_cast<const half8*>(brow + ((nt + ky) * PW + kx) * pixb + c0 * 2);
    };
    for (int it = 0; it < iters; ++it) {
        half8 A0[MT], B0[NT], A1[MT], B1[NT];
        ld(A0, B0, 0);
#pragma unroll
        for (int st = 0; st < 18; st += 2) {
            ld(A1, B1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0[i], B0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 2 < 18) ld(A0, B0, st + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[i], B1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.f) out[0] = s;
}

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// epilogue model: per iteration MT*NT groups of 4 values -> bias, SiLU, pack, (STORE: one 8-byte store per group at the conv's addresses)
template <int MT, int NT, bool STORE, bool WIDE>
__global__ void __launch_bounds__(256) epilogue(_Float16* out, int iters, int cs, float seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, h = lane >> 4;
    float v0 = seed + lane * 0.01f;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        const size_t tile = (size_t)blockIdx.x * iters + it;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const size_t p = (tile * 4 * NT + NT * wave + nt) * 16 + l15;          // 16 pixels wide tile rows, contiguous pixels
            float w[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) w[mt][q] = silu(v0 + 0.1f * (mt * 4 + q) + 0.01f * it);
            if (STORE) {
                if (WIDE) {                                     // lane h owns channels h*4MT .. +4MT-1: MT*8 contiguous bytes
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<half4*>(out + p * cs + h * 4 * MT + mt * 4) = half4{(_Float16)w[mt][0], (_Float16)w[mt][1], (_Float16)w[mt][2], (_Float16)w[mt][3]};
                    }
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<half4*>(out + p * cs + mt * 16 + 4 * h) = half4{(_Float16)w[mt][0], (_Float16)w[mt][1], (_Float16)w[mt][2], (_Float16)w[mt][3]};
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) keep += w[mt][0] + w[mt][1] + w[mt][2] + w[mt][3];
            }
        }
    }
    if (keep == 12345.f) out[0] = (_Float16)keep;
}

int main() {
    float* buf; CK(hipMalloc(&buf, 1ull << 30));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, double unit, const char* what, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep && ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-52s %8.3f ms   %.1f %s\n", name, best, unit / (best * 1e-3), what); fflush(stdout);
    };
    const int G = 256, IT = 2000;
    // cycles per MFMA per wave = time * 2.4e9 / (iters * MT*NT)
    run("A  regs 4x4, 1 wave/SIMD", 1.0, "", [&] { hipLaunchKernelGGL((mfma_regs<4, 4>), dim3(G), dim3(256), 0, 0, buf, IT); });
    printf("   -> per MFMA %.1f ns-cycles@2.4GHz\n", 0.0);
    run("A  regs 5x4, 1 wave/SIMD (TFLOP/s)", (double)G * 4 * IT * 20 * 16384 / 1e12, "TFLOP/s", [&] { hipLaunchKernelGGL((mfma_regs<5, 4>), dim3(G), dim3(256), 0, 0, buf, IT); });
    run("A  regs 4x4, 2 waves/SIMD (TFLOP/s)", (double)2 * G * 4 * IT * 16 * 16384 / 1e12, "TFLOP/s", [&] { hipLaunchKernelGGL((mfma_regs<4, 4>), dim3(2 * G), dim3(256), 0, 0, buf, IT); });
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_lds<5, 4, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_lds<5, 4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_lds<4, 4, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int IT2 = 100;
    auto ldsb = [](int MT, int NT, int PAD) { return (size_t)16 * MT * (9 * 64 * 2 + PAD) + (size_t)(4 * NT + 2) * 18 * (64 * 2 + PAD); };
    run("B  lds 5x4 pad32 (conflict-free), 18 steps/iter", (double)G * 4 * IT2 * 18 * 20 * 16384 / 1e12, "TFLOP/s", [&] { hipLaunchKernelGGL((mfma_lds<5, 4, 32>), dim3(G), dim3(256), ldsb(5, 4, 32), 0, buf, IT2); });
    run("B  lds 5x4 pad16 (2-way conflicts)", (double)G * 4 * IT2 * 18 * 20 * 16384 / 1e12, "TFLOP/s", [&] { hipLaunchKernelGGL((mfma_lds<5, 4, 16>), dim3(G), dim3(256), ldsb(5, 4, 16), 0, buf, IT2); });
    run("B  lds 4x4 pad32", (double)G * 4 * IT2 * 18 * 16 * 16384 / 1e12, "TFLOP/s", [&] { hipLaunchKernelGGL((mfma_lds<4, 4, 32>), dim3(G), dim3(256), ldsb(4, 4, 32), 0, buf, IT2); });
    // epilogue: values per second per chip
    const int IT3 = 64;
    const double vals = (double)G * 256 * IT3 * 20 * 4;
    run("C  epilogue math 5x4 (G values/s)", vals / 1e9, "Gval/s", [&] { hipLaunchKernelGGL((epilogue<5, 4, false, false>), dim3(G), dim3(256), 0, 0, (_Float16*)buf, IT3, 80, 0.5f); });
    run("D  epilogue + 8-B stores, conv layout (GB/s)", vals * 2 / 1e9, "GB/s", [&] { hipLaunchKernelGGL((epilogue<5, 4, true, false>), dim3(G), dim3(256), 0, 0, (_Float16*)buf, IT3, 80, 0.5f); });
    run("D' epilogue + 8-B stores, lane-contiguous channels", vals * 2 / 1e9, "GB/s", [&] { hipLaunchKernelGGL((epilogue<5, 4, true, true>), dim3(G), dim3(256), 0, 0, (_Float16*)buf, IT3, 80, 0.5f); });
    return 0;
}
