// VALU issue-rate calibration on gfx950: cycles per wave64 instruction on one SIMD for the instruction kinds the lane front
// end is made of, at 1 / 2 / 4 waves per SIMD (8 independent registers per wave, so no dependency stalls).
// build: hipcc --offload-arch=gfx950 -O3 tools/wvalu.hip -o tools/wvalu ; run: tools/wvalu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int ITERS = 512;          // x 8 registers x 4 = 16384 instructions per wave

// BODY(i) is one asm statement on register r[i]; a, b are loop-invariant inputs
#define KERNEL(NAME, ASM)                                                                                              \
    __global__ void __launch_bounds__(1024) NAME(unsigned* out, unsigned long long* cyc, unsigned seed) {              \
        unsigned r[8];                                                                                                 \
        unsigned a = seed * 2654435761u + threadIdx.x, b = seed ^ (threadIdx.x * 40503u);                              \
        float fa = __uint_as_float((a & 0x007FFFFFu) | 0x3F800000u), fb = 1.0f;                                        \
        (void)fa, (void)fb; int sc = (int)seed; (void)sc; const unsigned long long msk = 0x5555555555555555ull ^ seed; (void)msk; __shared__ unsigned lds_[16 * 256]; if (threadIdx.x < 16) lds_[threadIdx.x] = 0;                                                                                            \
        for (int i = 0; i < 8; ++i) r[i] = a + i * 77u;                                                                \
        unsigned long long r2[8];                                                                                      \
        for (int i = 0; i < 8; ++i) r2[i] = a + i;                                                                     \
        (void)r2;                                                                                                      \
        __builtin_amdgcn_s_barrier();                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < ITERS; ++it) {                                                                           \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM; }                                                 \
            }                                                                                                          \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        unsigned s = 0;                                                                                                \
        for (int i = 0; i < 8; ++i) s ^= r[i] ^ (unsigned)r2[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                     \
    }

KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(fb), "v"(fa)))
KERNEL(k_add_f32, asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(fa)))
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(fb)))
KERNEL(k_add_f32_abs, asm volatile("v_add_f32 %0, |%0|, |%1|" : "+v"(r[i]) : "v"(fa)))
KERNEL(k_max3_f32, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(fa), "v"(fb)))
KERNEL(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r2[i]) : "v"(r2[(i + 1) & 7])))
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_alignbyte, asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(r[i]) : "v"(a)))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc"))
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_mad_u32_u16, asm volatile("v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_dot4_u8, asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_dot2_u16, asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_sad_u8, asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_sub_i16, asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_max_i16, asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_mul_lo_u16, asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_mad_u16, asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_pk_lshl_b16, asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(r[i])))
KERNEL(k_pk_add_f16, asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_fma_f16, asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_mov_dpp, asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i])))
KERNEL(k_add_dpp, asm volatile("v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_mov_sdwa, asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(r[i]) : "v"(a)))
KERNEL(k_add_sdwa, asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(r[i]) : "v"(a)))
KERNEL(k_cvt_ubyte1, asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i])))
KERNEL(k_cvt_pk_u8_f32, asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(r[i]) : "v"(fa)))
KERNEL(k_cvt_u32_f32, asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r[i])))
KERNEL(k_med3_i32, asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_min3_u32, asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_add3_u32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_sub_u16_clamp, asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_min_u16, asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_pk_ashr_i16, asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(r[i])))

KERNEL(k_or_b32, asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_xor_b32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_lshlrev, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i])))
KERNEL(k_lshrrev, asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i])))
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_min_i32, asm volatile("v_min_i32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_mul_u32_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_bfe_u32, asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r[i])))
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(fa)))
KERNEL(k_fmac_f32, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(fa), "v"(fb)))
KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL(k_cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[i])))
KERNEL(k_cndmask_s, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(msk)))
KERNEL(k_cmp_gt, asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc"))
KERNEL(k_and_lit, asm volatile("v_and_b32 %0, 0x00ff00ff, %0" : "+v"(r[i])))
KERNEL(k_pk_add_lit, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "s"(0x00800080u)))
KERNEL(k_add_f32_sdwa, asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(r[i]) : "v"(fa)))
KERNEL(k_ds_add, asm volatile("ds_add_u32 %0, %1" : : "v"((r[i] & 0x3FCu) + ((threadIdx.x & 15u) << 10)), "v"(1u) : "memory"))

KERNEL(k_pk_add_dep1, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i & 0]) : "v"(a)))
KERNEL(k_pk_add_dep2, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i & 1]) : "v"(a)))
KERNEL(k_pk_add_dep4, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i & 3]) : "v"(a)))
KERNEL(k_perm_dep1, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i & 0]) : "v"(a), "v"(b)))
KERNEL(k_perm_dep2, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i & 1]) : "v"(a), "v"(b)))
KERNEL(k_add_u32_dep1, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i & 0]) : "v"(a)))
KERNEL(k_add_u32_dep2, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i & 1]) : "v"(a)))
KERNEL(k_mix_pk_perm, asm volatile("v_pk_add_u16 %0, %0, %1\n\tv_perm_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %1, %0, %2\n\tv_pk_sub_i16 %0, %0, %1" : "+v"(r[i & 1]) : "v"(a), "v"(b)))

KERNEL(k_mix4_salu1, asm volatile("v_pk_add_u16 %[r], %[r], %[a]\n\tv_perm_b32 %[r], %[r], %[a], %[b]\n\tv_bfi_b32 %[r], %[a], %[r], %[b]\n\tv_pk_sub_i16 %[r], %[r], %[a]\n\ts_add_i32 %[sc], %[sc], 1" : [r] "+v"(r[i & 1]), [sc] "+s"(sc) : [a] "v"(a), [b] "v"(b)))
KERNEL(k_mix4_salu2, asm volatile("v_pk_add_u16 %[r], %[r], %[a]\n\ts_cmp_lt_i32 %[sc], 77\n\tv_perm_b32 %[r], %[r], %[a], %[b]\n\tv_bfi_b32 %[r], %[a], %[r], %[b]\n\ts_cselect_b32 %[sc], %[sc], 5\n\tv_pk_sub_i16 %[r], %[r], %[a]" : [r] "+v"(r[i & 1]), [sc] "+s"(sc) : [a] "v"(a), [b] "v"(b) : "scc"))
KERNEL(k_mix4_cnd, asm volatile("v_pk_add_u16 %[r], %[r], %[a]\n\ts_cmp_lt_i32 %[sc], 77\n\tv_perm_b32 %[r], %[r], %[a], %[b]\n\ts_cselect_b64 vcc, -1, 0\n\tv_bfi_b32 %[r], %[a], %[r], %[b]\n\tv_cndmask_b32 %[r], %[r], %[a], vcc" : [r] "+v"(r[i & 1]), [sc] "+s"(sc) : [a] "v"(a), [b] "v"(b) : "scc", "vcc"))
KERNEL(k_mix4_dpp, asm volatile("v_pk_add_u16 %0, %0, %1\n\tv_perm_b32 %0, %0, %1, %2\n\tv_bfi_b32 %0, %1, %0, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i & 1]) : "v"(a), "v"(b)))
KERNEL(k_mix4_3regs, asm volatile("v_pk_add_u16 %0, %3, %1\n\tv_perm_b32 %0, %0, %4, %2\n\tv_bfi_b32 %0, %1, %0, %5\n\tv_pk_sub_i16 %0, %0, %3" : "+v"(r[i & 1]) : "v"(a), "v"(b), "v"(r[(i + 2) & 7]), "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7])))


// the same instruction in a LONG loop body (768 instructions = 6 KB, like the lane front end's unrolled row loop): is the
// issue rate of a small loop kept when the code no longer sits in the waves' instruction buffers?
#define KERNEL_BIG(NAME, ASM)                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(unsigned* out, unsigned long long* cyc, unsigned seed) {              \
        unsigned r[8];                                                                                                 \
        unsigned a = seed * 2654435761u + threadIdx.x, b = seed ^ (threadIdx.x * 40503u);                              \
        for (int i = 0; i < 8; ++i) r[i] = a + i * 77u;                                                                \
        __builtin_amdgcn_s_barrier();                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < ITERS * 32 / 768; ++it) {                                                                \
            _Pragma("unroll") for (int u = 0; u < 96; ++u) {                                                           \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM; }                                                 \
            }                                                                                                          \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        unsigned s = 0;                                                                                                \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + b;                                                            \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) * 768 * (ITERS * 32 / 768) / (ITERS * 32);                     \
    }
KERNEL_BIG(k_big_pk_add, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL_BIG(k_big_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
KERNEL_BIG(k_big_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))

// semantics probe: what does v_cvt_pk_u8_f32 do with fractions, and v_cvt_u32_f32
__global__ void probe_kernel(float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned d = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(in[i]));
    out[i] = d;
}

struct K { const char* name; void (*fn)(unsigned*, unsigned long long*, unsigned); };
#define E(N) {#N, N}

int main(int argc, char** argv) {
    K ks[] = {E(k_fma_f32), E(k_add_f32), E(k_mul_f32), E(k_add_f32_abs), E(k_max3_f32), E(k_pk_fma_f32), E(k_add_u32), E(k_and_b32),
              E(k_lshl_or), E(k_and_or), E(k_bfi), E(k_perm), E(k_alignbyte), E(k_cndmask), E(k_mad_u24), E(k_mad_u32_u16),
              E(k_dot4_u8), E(k_dot2_u16), E(k_sad_u8), E(k_pk_add_u16), E(k_pk_sub_i16), E(k_pk_max_i16), E(k_pk_mul_lo_u16),
              E(k_pk_mad_u16), E(k_pk_lshl_b16), E(k_pk_add_f16), E(k_pk_fma_f16), E(k_mov_dpp), E(k_add_dpp), E(k_mov_sdwa),
              E(k_add_sdwa), E(k_cvt_ubyte1), E(k_cvt_pk_u8_f32), E(k_cvt_u32_f32), E(k_med3_i32), E(k_min3_u32), E(k_add3_u32),
              E(k_lshl_add), E(k_pk_sub_u16_clamp), E(k_pk_min_u16), E(k_pk_ashr_i16), E(k_or_b32), E(k_xor_b32), E(k_lshlrev), E(k_lshrrev), E(k_sub_u32), E(k_max_u32), E(k_min_i32), E(k_mul_u32_u24), E(k_mul_lo_u32), E(k_bfe_u32), E(k_max_f32), E(k_fmac_f32), E(k_mov_b32), E(k_cvt_f32_u32), E(k_cndmask_s), E(k_cmp_gt), E(k_and_lit), E(k_pk_add_lit), E(k_add_f32_sdwa), E(k_ds_add), E(k_pk_add_dep1), E(k_pk_add_dep2), E(k_pk_add_dep4), E(k_perm_dep1), E(k_perm_dep2), E(k_add_u32_dep1), E(k_add_u32_dep2), E(k_mix_pk_perm), E(k_mix4_salu1), E(k_mix4_salu2), E(k_mix4_cnd), E(k_mix4_dpp), E(k_mix4_3regs), E(k_big_pk_add), E(k_big_add_u32), E(k_big_perm)};
    const int nbmax = 512;
    unsigned* out;
    unsigned long long* cyc;
    hipMalloc(&out, nbmax * 1024 * 4);
    hipMalloc(&cyc, nbmax * 16 * 8);
    std::vector<unsigned long long> h(nbmax * 16);
    printf("%-22s %10s %10s %10s   (cycles per wave-instruction per SIMD = wave cycles / instructions / waves per SIMD)\n", "instruction",
           "1 w/SIMD", "2 w/SIMD", "4 w/SIMD"); printf("(4th column: 8 w/SIMD = two 1024-thread workgroups per CU)\n");
    for (auto& k : ks) {
        if (argc > 1 && !strstr(k.name, argv[1])) continue;
        printf("%-22s", k.name);
        for (int wps : {1, 2, 4, 6, 8}) {
            const int threads = wps == 8 ? 1024 : (wps == 6 ? 768 : 256 * wps);
            const int nb = wps >= 6 ? 512 : 256;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.fn, dim3(nb), dim3(threads), 0, 0, out, cyc, 12345u + rep);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), cyc, nb * (threads / 64) * 8, hipMemcpyDeviceToHost);
            double s = 0;
            const int nw = nb * (threads / 64);
            for (int i = 0; i < nw; ++i) s += (double)h[i];
            const double per = s / nw / (ITERS * 32.0) / wps;
            printf(" %10.2f", per);
        }
        printf("\n");
    }
    // probe
    float hin[16] = {0.0f, 0.25f, 0.5f, 0.75f, 1.0f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 254.5f, 255.4f, 255.5f, 256.0f, 300.0f, -1.0f};
    float* din;
    unsigned* dout;
    hipMalloc(&din, 64);
    hipMalloc(&dout, 64);
    hipMemcpy(din, hin, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, din, dout, 16);
    unsigned ho[16];
    hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_u8_f32:");
    for (int i = 0; i < 16; ++i) printf(" %g->%u", hin[i], ho[i]);
    printf("\n");
    return 0;
}
