// Dependent-issue latency of f64 VALU ops on one wave per SIMD (gfx950), via inline asm chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void k(double* out, unsigned long long* cyc, double seed)
{
    double x = seed + threadIdx.x * 1e-9, y = 0.999999, z = 1e-9;
    unsigned long long t0, t1;
    // dependent v_fma_f64
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(x) : "v"(y), "v"(z));
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x) : "memory");
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // dependent v_mul_f64
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(x) : "v"(y));
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x) : "memory");
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // independent v_fma_f64 (4 chains)
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y), "v"(z));
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // dependent v_rsq_f64
    double r = x * x + 2.0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile(REP64("v_rsq_f64 %0, %0\n") : "+v"(r));
    asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // dependent v_cndmask pair + v_cmp_class (as in safe_rsqrt)
    out[threadIdx.x] = x + a + b + c + d + r;
}
int main()
{
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&cyc, 64);
    for (int waves = 1; waves <= 4; waves *= 4) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.5);
        unsigned long long h[4]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d wave(s)/CU: dep v_fma_f64 %.1f | dep v_mul_f64 %.1f | 4 indep fma chains %.1f per op | dep v_rsq_f64 %.1f cycles\n",
               waves, h[0] / 64.0, h[1] / 64.0, h[2] / 256.0, h[3] / 64.0);
    }
    return 0;
}
