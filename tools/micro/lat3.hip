// Does the f64 MFMA rate hold when all 4 SIMDs of a CU issue it at once?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out, unsigned long long* cyc, double seed, int active_waves)
{
    const int w = threadIdx.x >> 6;
    double4_t c = { seed, 1.0, 2.0, 3.0 }, d = { 1.0, seed, 2.0, 3.0 };
    double y = 0.999 + threadIdx.x * 1e-6;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (w < active_waves) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, d, 0, 0, 0);
        }
    }
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(c), "v"(d) : "memory");
    if ((threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
    out[threadIdx.x] = c[0] + d[1];
}
int main()
{
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 512 * 8); (void)hipMalloc(&cyc, 64);
    for (int aw = 1; aw <= 8; aw *= 2) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * (aw < 4 ? 4 : aw)), 0, 0, out, cyc, 1.5, aw);
        unsigned long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d wave(s) issuing 64 independent-pair f64 MFMAs: wave0 %.1f cycles per MFMA", aw, h[0] / 64.0);
        if (aw >= 4) printf(", wave3 %.1f", h[3] / 64.0);
        printf("\n");
    }
    return 0;
}
