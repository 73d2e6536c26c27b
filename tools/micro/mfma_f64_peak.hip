// Measured v_mfma_f64_16x16x4_f64 rate on MI355X (gfx950): single wave with 1/2/4/8 independent
// accumulators, and the whole chip (256 CUs x 8 waves, 8 accumulators each) for the achievable peak.
// bench.py prices the MFMA-bound kernels against the chip-wide number printed here
// (MI355X_MICROARCH.md has no f64 row).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_wave(double* out, unsigned long long* cyc, double seed)
{
    double4_t c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (double4_t){ seed + i, 1.0, 2.0, 3.0 };
    double y = 0.999 + threadIdx.x * 1e-6;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int it = 0; it < 64 / NACC; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0];
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_chip(double* out, double seed, int iters)
{
    double4_t c[8];
    for (int i = 0; i < 8; ++i) c[i] = (double4_t){ seed + i, 1.0, 2.0, 3.0 };
    double y = 0.999 + threadIdx.x * 1e-6, z = 1.001 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main()
{
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 2048 * 512 * 8); (void)hipMalloc(&cyc, 64);
    unsigned long long h;
    hipLaunchKernelGGL(k_wave<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("1 accumulator : %.1f cycles/MFMA\n", h / 64.0);
    hipLaunchKernelGGL(k_wave<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("2 accumulators: %.1f cycles/MFMA\n", h / 64.0);
    hipLaunchKernelGGL(k_wave<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("4 accumulators: %.1f cycles/MFMA\n", h / 64.0);
    hipLaunchKernelGGL(k_wave<8>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("8 accumulators: %.1f cycles/MFMA\n", h / 64.0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int blocks : { 256, 512, 1024 }) {
        hipLaunchKernelGGL(k_chip, dim3(blocks), dim3(512), 0, 0, out, 1.5, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_chip, dim3(blocks), dim3(512), 0, 0, out, 1.5, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 8 /*waves*/ * iters * 8.0 * 2048.0;
        printf("chip-wide, %4d workgroups x 8 waves: %.1f TFLOP/s f64 MFMA (%.2f ms)\n", blocks, flops / ms * 1e-9, ms);
    }
    return 0;
}
