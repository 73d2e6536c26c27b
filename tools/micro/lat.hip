// Latency microbenchmarks on one wave (gfx950): dependent f64 FMA chain, v_rsq_f64, LDS read,
// f64 MFMA dependent chain, s_barrier with 4 waves.  Prints cycles per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_lat(double* out, unsigned long long* cyc, double seed)
{
    __shared__ double lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += blockDim.x) lds[i] = (double)((i * 7 + 3) & 1023);
    __syncthreads();
    unsigned long long t0, t1;
    double x = seed + tid * 1e-9, y = 0.999999;
    // 1. dependent fma chain
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 256; ++i) x = fma(x, y, 1e-9);
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[0] = t1 - t0;
    // 2. independent fma (4 chains)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 64; ++i) { a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9); }
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[1] = t1 - t0;
    x = a0 + a1 + a2 + a3;
    // 3. dependent rsq chain
    double r = fabs(x) + 2.0;
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rsq(r) + 2.0;
    asm volatile("" :: "v"(r));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[2] = t1 - t0;
    // 4. dependent LDS read chain (pointer chase)
    int idx = tid & 1023;
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 64; ++i) idx = (int)lds[idx];
    asm volatile("" :: "v"(idx));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[3] = t1 - t0;
    // 5. dependent MFMA f64 chain
    double4_t c = { x, r, 1.0, 2.0 };
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 32; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
    asm volatile("" :: "v"(c));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[4] = t1 - t0;
    // 6. barriers
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 32; ++i) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[5] = t1 - t0;
    // 7. MFMA result -> VALU use latency: mfma then dependent fma on result, chain
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 32; ++i) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0); c[0] = fma(c[0], y, 1e-9); }
    asm volatile("" :: "v"(c));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[6] = t1 - t0;
    // 8. LDS write then read same address by another lane after barrier
    t0 = __builtin_amdgcn_s_memtime();
    #pragma unroll
    for (int i = 0; i < 32; ++i) { lds[tid] = x; __syncthreads(); x = lds[(tid + 1) & 255] + 1.0; __syncthreads(); }
    asm volatile("" :: "v"(x));
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[7] = t1 - t0;
    out[tid] = x + r + idx + c[0] + c[1];
}
int main()
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(256), 0, 0, out, cyc, 1.5);
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("dep fma f64: %.1f cyc | 4-way indep fma: %.1f cyc/op | dep rsq_f64(+add): %.1f | LDS read dep: %.1f | dep mfma f64: %.1f | barrier(4 waves): %.1f | mfma->valu->mfma: %.1f | lds wr+bar+rd+bar: %.1f\n",
               h[0] / 256.0, h[1] / 256.0, h[2] / 64.0, h[3] / 64.0, h[4] / 32.0, h[5] / 32.0, h[6] / 32.0, h[7] / 32.0);
    }
    return 0;
}
