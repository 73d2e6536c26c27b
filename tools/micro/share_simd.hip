// What does an f64 VALU wave lose when ANOTHER wave of its SIMD streams f64 MFMAs?  (gfx950)
// One workgroup of 8 waves; waves w and w + 4 share SIMD w (checked through HW_ID).  Wave 0 times a chain of
// dependent v_fma_f64 and then a stream of independent ones; wave 4 meanwhile does nothing / streams independent
// v_mfma_f64_16x16x4_f64 / streams f64 VALU work; with and without s_setprio 3 on wave 0.  Decides whether a
// latency-bound f64 kernel (the dataflow Cholesky's pivot waves) can share compute units with an MFMA-bound one
// (the rank-k update): DESIGN.md section 4.8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, unsigned* hwid, double seed, int partner, int prio,
                                         volatile int* stop)
{
    const int w = threadIdx.x >> 6;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) hwid[w] = id;
    __shared__ int s_stop;
    if (threadIdx.x == 0) s_stop = 0;
    __syncthreads();
    double r = 0.0;
    if (w == 0) {
        if (prio) asm volatile("s_setprio 3");
        // let the partner get going
        for (int i = 0; i < 2000; ++i) asm volatile("s_nop 15");
        double x = seed + threadIdx.x * 1e-9, a = 0.999999, b = 1e-7;
        unsigned long long t0, t1, t2;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
        for (int i = 0; i < 256; ++i)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x) : "memory");
        double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3, y4 = x + 4, y5 = x + 5, y6 = x + 6, y7 = x + 7;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2) : "v"(y0), "v"(y7) : "memory");
        r = x + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
        if ((threadIdx.x & 63) == 0) {
            cyc[0] = t1 - t0;
            cyc[1] = t2 - t1;
            s_stop = 1;
        }
    } else if (w == 4 && partner) {
        double4_t c0 = { seed, 1, 2, 3 }, c1 = { 1, seed, 2, 3 }, c2 = { 2, 1, seed, 3 }, c3 = { 3, 2, 1, seed };
        double y = 0.999 + threadIdx.x * 1e-6, z0 = y, z1 = y + 1, z2 = y + 2, z3 = y + 3;
        unsigned long long t0, t1;
        unsigned long long n = 0;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        while (!*(volatile int*)&s_stop && n < 100000) {
            if (partner == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c3, 0, 0, 0);
                }
                n += 16;
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                                 : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(y));
                n += 64;
            }
        }
        asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(c0), "v"(c1), "v"(c2), "v"(c3) : "memory");
        r = c0[0] + c1[1] + c2[2] + c3[3] + z0 + z1 + z2 + z3;
        if ((threadIdx.x & 63) == 0) {
            cyc[2] = t1 - t0;
            cyc[3] = n;
        }
    }
    out[threadIdx.x] = r;
    (void)stop;
}

int main()
{
    double* out; unsigned long long* cyc; unsigned* hwid; int* stop;
    (void)hipMalloc(&out, 512 * 8); (void)hipMalloc(&cyc, 64); (void)hipMalloc(&hwid, 64); (void)hipMalloc(&stop, 4);
    const char* names[3] = { "idle", "f64 MFMA stream", "f64 VALU stream" };
    for (int prio = 0; prio < 2; ++prio)
        for (int partner = 0; partner < 3; ++partner) {
            (void)hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, hwid, 1.5, partner, prio, stop);
            unsigned long long h[4]; unsigned id[8];
            if (hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
            (void)hipMemcpy(id, hwid, sizeof(id), hipMemcpyDeviceToHost);
            printf("partner on the SIMD: %-16s prio %d | dependent v_fma_f64 %.1f cycles each, independent %.1f | partner: %.1f cycles per instruction"
                   " | SIMD of wave 0 / wave 4: %u / %u\n",
                   names[partner], prio * 3, h[0] / 256.0, h[1] / 256.0, h[3] ? (double)h[2] / h[3] : 0.0, (id[0] >> 4) & 3, (id[4] >> 4) & 3);
        }
    return 0;
}
