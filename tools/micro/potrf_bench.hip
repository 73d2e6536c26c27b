// Stand-alone timing of the 64x64 in-LDS Cholesky used by the panel kernel (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define STAMP(x)
#define RSTAMP(x)
namespace vmm { struct LmCtl { int done, lin_fail; }; constexpr int kNB = 64; }
#define VMM_POTRF_ONLY
#include "../../visual_marker_mapping_amd/csrc/potrf64.inc"

__global__ __launch_bounds__(256) void k(const double* S, double* Lout, unsigned long long* cyc)
{
    __shared__ double A[64 * vmm::kLd];
    __shared__ __attribute__((aligned(16))) double At[64 * vmm::kLdT];
    __shared__ double invd[64];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 4096; idx += 256) { int r = idx >> 6, c = idx & 63; A[r * vmm::kLd + c] = c <= r ? S[idx] : 0.0; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    bool ok = vmm::potrf64_lds(A, At, invd);
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int idx = tid; idx < 4096; idx += 256) { int r = idx >> 6, c = idx & 63; Lout[idx] = At[c * vmm::kLdT + r]; }
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = ok; }
}
int main()
{
    std::vector<double> B(4096), S(4096, 0.0), L(4096);
    unsigned s = 12345;
    for (auto& v : B) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0 - 0.5; }
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double a = 0; for (int k = 0; k < 64; ++k) a += B[i*64+k]*B[j*64+k]; S[i*64+j] = a + (i==j ? 8.0 : 0.0); }
    double *dS, *dL; unsigned long long* dc;
    (void)hipMalloc(&dS, 4096*8); (void)hipMalloc(&dL, 4096*8); (void)hipMalloc(&dc, 16);
    (void)hipMemcpy(dS, S.data(), 4096*8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dS, dL, dc);
        unsigned long long c[2]; (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(L.data(), dL, 4096*8, hipMemcpyDeviceToHost);
        double err = 0;
        for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) { double a = 0; for (int k = 0; k <= j; ++k) a += L[i*64+k]*L[j*64+k]; err = fmax(err, fabs(a - S[i*64+j])); }
        printf("potrf64: %llu cycles (%.2f us @2.4GHz), ok=%llu, max |LL^T - S| = %.3e\n", c[0], c[0] / 2400.0, c[1], err);
    }
    return 0;
}
