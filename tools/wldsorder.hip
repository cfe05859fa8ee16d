// Does one ds_add_rtn_u32 wave instruction serve lanes that hit the SAME LDS address in increasing lane order (each lane getting
// the running value)?  houghp_shard could then vote several batch points per instruction (point b in a lower lane than point
// b + 1 of the same theta row).  Patterns: all 64 lanes one address; lane pairs (l, l + 32); groups of four (l, l+16, l+32, l+48);
// random groups.  Reports the number of violations of "returned value == number of lower lanes with the same address" over many
// trials, also with a second wave of the workgroup hammering other addresses of the same banks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const int* addr, int* out, int n_trials, int noise) {
    __shared__ unsigned cell[256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = 0; t < n_trials; ++t) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) cell[i] = 0;
        __syncthreads();
        if (wv == 0) {
            const int a = addr[t * 64 + lane];
            const unsigned old = atomicAdd(&cell[a], 1u);
            out[t * 64 + lane] = (int)old;
        } else if (noise) {
            for (int q = 0; q < 8; ++q) atomicAdd(&cell[128 + ((lane * 7 + q * 13) & 127)], 1u);
        }
        __syncthreads();
    }
}

int main() {
    const int T = 4096;
    std::vector<int> addr(T * 64), out(T * 64);
    srand(7);
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            int a;
            switch (t & 3) {
                case 0: a = 5; break;
                case 1: a = l & 31; break;
                case 2: a = l & 15; break;
                default: a = rand() % 24; break;
            }
            addr[t * 64 + l] = a;
        }
    int *da, *dout;
    hipMalloc(&da, addr.size() * 4), hipMalloc(&dout, out.size() * 4);
    hipMemcpy(da, addr.data(), addr.size() * 4, hipMemcpyHostToDevice);
    for (int noise = 0; noise < 2; ++noise) {
        hipLaunchKernelGGL(k, dim3(64), dim3(128), 0, 0, da, dout, T, noise);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        long bad[4] = {0, 0, 0, 0};
        for (int t = 0; t < T; ++t)
            for (int l = 0; l < 64; ++l) {
                int want = 0;
                for (int m = 0; m < l; ++m) want += addr[t * 64 + m] == addr[t * 64 + l];
                if (out[t * 64 + l] != want) ++bad[t & 3];
            }
        printf("noise %d: violations all-same %ld, pairs %ld, quads %ld, random %ld (of %d lanes each)\n", noise, bad[0], bad[1], bad[2], bad[3], T / 4 * 64);
    }
    return 0;
}
