// Checks and times the DPP pivot-block primitives of csrc/chol_dpp.hpp on one wave: W x W Cholesky with a matrix row per
// lane, then the scaling of a 64-row panel, against the same operations done on the host in the same order (fma).
// build: hipcc -O3 --offload-arch=gfx950 -I visual_marker_mapping_amd/csrc tools/micro/dpp_chol.hip -o /tmp/dpp_chol
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "chol_dpp.hpp"

template <int W>
__global__ void k_test(const double* __restrict__ A, const double* __restrict__ X, double* __restrict__ Lout,
                       double* __restrict__ Xout, unsigned long long* cyc, int* okout)
{
    const int lane = threadIdx.x & 63;
    double a[W], inv[W], x[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        a[c] = A[((lane & 15) % W) * W + c];
        x[c] = X[lane * W + c];
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < W; ++c)
        asm volatile("" : "+v"(a[c]), "+v"(x[c]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    vmm::dpp::chol_rows<W>(a, inv, ok);
#pragma unroll
    for (int c = 0; c < W; ++c)
        asm volatile("" : "+v"(a[c]), "+v"(inv[c]), "+v"(x[c]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    vmm::dpp::scale_row<W>(x, a, inv);
#pragma unroll
    for (int c = 0; c < W; ++c)
        asm volatile("" : "+v"(x[c]));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        cyc[0] = t1 - t0;
        cyc[1] = t2 - t1;
        okout[0] = ok ? 1 : 0;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {
        Lout[threadIdx.x * W + c] = a[c];
        Xout[threadIdx.x * W + c] = x[c];
    }
}

static double rsqrt_host(double v)
{
    return 1.0 / std::sqrt(v);
}

template <int W>
int run()
{
    std::vector<double> A(W * W), B(W * W), X(256 * W);
    srand(7 + W);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < W; ++j) {
            double s = (i == j) ? W * 0.5 : 0.0;
            for (int k = 0; k < W; ++k) s += B[i * W + k] * B[j * W + k];
            A[i * W + j] = s;
        }
    for (auto& v : X) v = rand() / (double)RAND_MAX - 0.5;
    // host: same right-looking order
    std::vector<double> L = A, inv(W);
    for (int j = 0; j < W; ++j) {
        const double r = rsqrt_host(L[j * W + j]);
        inv[j] = r;
        for (int i = j; i < W; ++i) L[i * W + j] *= r;
        for (int c = j + 1; c < W; ++c)
            for (int i = c; i < W; ++i) L[i * W + c] = std::fma(-L[c * W + j], L[i * W + j], L[i * W + c]);
    }
    std::vector<double> Xh = X;
    for (int row = 0; row < 256; ++row)
        for (int q = 0; q < W; ++q) {
            Xh[row * W + q] *= inv[q];
            for (int c = q + 1; c < W; ++c)
                Xh[row * W + c] = std::fma(-L[c * W + q], Xh[row * W + q], Xh[row * W + c]);
        }
    double *dA, *dX, *dL, *dXo;
    unsigned long long* dc;
    int* dok;
    (void)hipMalloc(&dA, A.size() * 8);
    (void)hipMalloc(&dX, X.size() * 8);
    (void)hipMalloc(&dL, 256 * W * 8);
    (void)hipMalloc(&dXo, 256 * W * 8);
    (void)hipMalloc(&dc, 16);
    (void)hipMalloc(&dok, 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    int bad = 0;
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(k_test<W>, dim3(1), dim3(64 * waves), 0, 0, dA, dX, dL, dXo, dc, dok);
        std::vector<double> Lg(256 * W), Xg(256 * W);
        unsigned long long c[2];
        int ok;
        (void)hipMemcpy(Lg.data(), dL, Lg.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(Xg.data(), dXo, Xg.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
        double eL = 0, eX = 0;
        for (int t = 0; t < 64 * waves; ++t) {
            const int i = (t & 15) % W;
            for (int cc = 0; cc <= i; ++cc) eL = std::fmax(eL, std::fabs(Lg[t * W + cc] - L[i * W + cc]));
            for (int cc = 0; cc < W; ++cc) eX = std::fmax(eX, std::fabs(Xg[t * W + cc] - Xh[(t & 63) * W + cc]));
        }
        printf("W=%2d waves=%d: chol %llu cycles, scale %llu cycles, ok=%d, max|L-Lhost| %.3g, max|X-Xhost| %.3g\n", W, waves, c[0],
               c[1], ok, eL, eX);
        if (!(eL < 1e-13) || !(eX < 1e-12) || !ok) bad = 1;
    }
    return bad;
}

int main()
{
    int bad = run<8>();
    bad |= run<16>();
    printf(bad ? "FAILED\n" : "PASSED\n");
    return bad;
}
