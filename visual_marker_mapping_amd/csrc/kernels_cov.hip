// Covariance of the tag translations from the Schur factor (replaces the ceres::Covariance block of
// /root/reference/src/TagReconstructor.cpp:744-783): 3x3 diagonal blocks of (J^T J)^{-1} in tangent
// coordinates at the current state.
//
// With H = [H_EE H_EF; H_FE H_FF] (E eliminated, F kept), M_e = L_e L_e^T, Z = L_E^{-1} H_EF and
// S = H_FF - Z^T Z = L L^T (all produced by the iteration's own kernels run with scale = 1, D^2 = 0):
//   tags kept (F):        Cov_FF = S^{-1} = X^T X            with X = L^{-1}            (rhs = I)
//   tags eliminated (E):  Cov_jj = L_j^{-T} (I + U_j^T U_j) L_j^{-1}   with U = L^{-1} Z^T  (rhs = Z^T)
// so both cases are one blocked forward substitution L X = B with many right-hand sides followed by the
// 6x6 Gram matrices of 6-column groups of X.  The substitution walks the 64-row blocks of L: block row k
// is multiplied by the explicit inverse of its diagonal factor (already formed for the back-substitution
// chain), then subtracted from the rows below.  This runs once per reconstruction (final summary), so the
// tile kernels are plain LDS-tiled f64 FMA code, not MFMA.
#include "engine.hpp"

#include <hip/hip_runtime.h>

namespace vmm {

// active = has a non-zero Jacobian block; unit scaling; no damping, but unit diagonal for inactive poses
// (their rows and columns of H are zero) so that every block stays positive definite
__global__ void k_cov_prepare(int n_pose, const double* __restrict__ H, double* __restrict__ scale,
                              double* __restrict__ D2, int32_t* __restrict__ active)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pose)
        return;
    const double* Hp = H + 36 * (int64_t)p;
    const int a = (Hp[0] + Hp[7] + Hp[14]) > 0.0 ? 1 : 0;
    active[p] = a;
    for (int k = 0; k < 6; ++k) {
        scale[6 * (int64_t)p + k] = 1.0;
        D2[6 * (int64_t)p + k] = a ? 0.0 : 1.0;
    }
}

__global__ void k_cov_identity(double* __restrict__ B, int ldb, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        B[(int64_t)i * ldb + i] = 1.0;
}

// B[r][c] = Z[c][r]: 32x32 tiles through LDS
__global__ __launch_bounds__(256) void k_cov_transpose(const double* __restrict__ Z, int ldz, int k_dim, int n_red,
                                                        double* __restrict__ B, int ldb)
{
    __shared__ double t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;   // c: row of Z, r: column of Z
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        t[j][tx] = (c < k_dim && r < n_red) ? Z[(int64_t)c * ldz + r] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < n_red && c < k_dim)
            B[(int64_t)r * ldb + c] = t[tx][j];
    }
}

constexpr int kTa = 65, kTb = 68;

// acc(4x4 per thread) of C(64x64) = A(64x64, row-major, lda) * Bm(64 x 64, row-major, ldb)
__device__ __forceinline__ void tile_product(const double* __restrict__ A, int lda, const double* __restrict__ Bm,
                                             int ldb, double* As, double* Bs, double (&acc)[4][4])
{
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        As[r * kTa + c] = A[(int64_t)r * lda + c];
        Bs[r * kTb + c] = Bm[(int64_t)r * ldb + c];
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = 0.0;
    for (int m = 0; m < 64; ++m) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = As[(ty * 4 + i) * kTa + m];
            b[i] = Bs[m * kTb + tx * 4 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] += a[i] * b[j];
    }
}

// X_k = Linv_k B_k in place; one workgroup per 64-column chunk.  first_chunk: chunks below it are zero
// (identity right-hand side: X is lower triangular) and skipped by the launch.
__global__ __launch_bounds__(256) void k_trsm_diag(const double* __restrict__ Linv_k, double* __restrict__ B, int ldb,
                                                   int k, int first_chunk)
{
    __shared__ double As[64 * kTa], Bs[64 * kTb];
    const int c0 = (first_chunk + (int)blockIdx.x) * 64;
    double* Bk = B + (int64_t)k * 64 * ldb + c0;
    double acc[4][4];
    tile_product(Linv_k, 64, Bk, ldb, As, Bs, acc);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            Bk[(int64_t)(ty * 4 + i) * ldb + tx * 4 + j] = acc[i][j];
}

// B_i -= L_ik X_k for the block rows i > k; grid (column chunks, block rows below k)
__global__ __launch_bounds__(256) void k_trsm_update(const double* __restrict__ S, int ld, double* __restrict__ B,
                                                     int ldb, int k, int first_chunk)
{
    __shared__ double As[64 * kTa], Bs[64 * kTb];
    const int c0 = (first_chunk + (int)blockIdx.x) * 64;
    const int i = k + 1 + (int)blockIdx.y;
    const double* Lik = S + (int64_t)i * 64 * ld + (int64_t)k * 64;
    const double* Xk = B + (int64_t)k * 64 * ldb + c0;
    double* Bi = B + (int64_t)i * 64 * ldb + c0;
    double acc[4][4];
    tile_product(Lik, ld, Xk, ldb, As, Bs, acc);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            Bi[(int64_t)(ty * 4 + a) * ldb + tx * 4 + j] -= acc[a][j];
}

// 3x3 translation covariance of one tag from the 6 columns of X that belong to it.  One workgroup per tag.
// kept == true:  cov = (X^T X)[0:3, 0:3];  kept == false:  cov = (L_j^{-T} (I + X^T X) L_j^{-1})[0:3, 0:3].
__global__ __launch_bounds__(256) void k_cov_gram(const double* __restrict__ X, int ldb, int n_rows, int n_tags,
                                                   bool kept, const double* __restrict__ Le,
                                                   const int32_t* __restrict__ active_tag, double* __restrict__ cov)
{
    __shared__ double sh[21][256 + 1];
    const int j = blockIdx.x;
    const int tid = threadIdx.x;
    double g[21];
#pragma unroll
    for (int q = 0; q < 21; ++q)
        g[q] = 0.0;
    for (int r = tid; r < n_rows; r += 256) {
        double x[6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
            x[a] = X[(int64_t)r * ldb + 6 * j + a];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b)
                g[q++] += x[a] * x[b];
    }
#pragma unroll
    for (int q = 0; q < 21; ++q)
        sh[q][tid] = g[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int q = 0; q < 21; ++q)
                sh[q][tid] += sh[q][tid + s];
        __syncthreads();
    }
    if (tid != 0)
        return;
    double* out = cov + 9 * (int64_t)j;
    if (!active_tag[j]) {   // constant (origin) or residual-free block: Ceres reports zero covariance
        for (int q = 0; q < 9; ++q)
            out[q] = 0.0;
        return;
    }
    double G[6][6];
    int q = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b <= a; ++b) {
            G[a][b] = sh[q][0];
            G[b][a] = sh[q][0];
            ++q;
        }
    if (!kept) {
        // T = L_j^{-1} (lower), C = T^T (I + G) T
        const double* L = Le + 36 * (int64_t)j;
        double T[6][6];
        for (int c = 0; c < 6; ++c)
            for (int r = 0; r < 6; ++r) {
                if (r < c) {
                    T[r][c] = 0.0;
                    continue;
                }
                double s = (r == c) ? 1.0 : 0.0;
                for (int m = c; m < r; ++m)
                    s -= L[6 * r + m] * T[m][c];
                T[r][c] = s / L[6 * r + r];
            }
        for (int a = 0; a < 6; ++a)
            G[a][a] += 1.0;
        double GT[6][6];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
                double s = 0.0;
                for (int m = 0; m < 6; ++m)
                    s += G[a][m] * T[m][b];
                GT[a][b] = s;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0.0;
                for (int m = 0; m < 6; ++m)
                    s += T[m][a] * GT[m][b];
                out[3 * a + b] = s;
            }
        return;
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            out[3 * a + b] = G[a][b];
}

// ---- launchers -------------------------------------------------------------------------------------

void launch_cov_prepare(Engine& e)
{
    const int n_pose = e.n_cams + e.n_tags;
    // H_cam and H_tag are adjacent in the small buffer (cameras first), like scale / D2 / active
    hipLaunchKernelGGL(k_cov_prepare, dim3((n_pose + 255) / 256), dim3(256), 0, e.stream, n_pose, e.H_cam, e.scale,
                       e.D2, e.active);
}

// Forward substitution L X = B on n_blk block rows; B is [n_pad][ldb], n_chunks 64-column chunks.
void launch_cov_trsm(Engine& e, double* B, int ldb, int n_chunks, bool identity_rhs)
{
    for (int k = 0; k < e.n_blk; ++k) {
        // identity right-hand side: block row k is non-zero in chunks 0..k only, and chunk c is zero above row c
        const int chunks = identity_rhs ? std::min(n_chunks, k + 1) : n_chunks;
        hipLaunchKernelGGL(k_trsm_diag, dim3(chunks), dim3(256), 0, e.stream,
                           (const double*)(e.Linv + (size_t)k * 4096), B, ldb, k, 0);
        if (k + 1 < e.n_blk)
            hipLaunchKernelGGL(k_trsm_update, dim3(chunks, e.n_blk - 1 - k), dim3(256), 0, e.stream,
                               (const double*)e.S, e.ldz, B, ldb, k, 0);
    }
}

void launch_cov_rhs(Engine& e, double* B, int ldb, bool identity_rhs)
{
    if (identity_rhs) {
        hipLaunchKernelGGL(k_cov_identity, dim3((e.n_pad + 255) / 256), dim3(256), 0, e.stream, B, ldb, e.n_pad);
    } else {
        hipLaunchKernelGGL(k_cov_transpose, dim3((e.k_dim + 31) / 32, (e.n_red + 31) / 32), dim3(256), 0, e.stream,
                           (const double*)e.Z, e.ldz, e.k_dim, e.n_red, B, ldb);
    }
}

void launch_cov_gram(Engine& e, const double* X, int ldb, double* cov_dev)
{
    const bool kept = e.elim_cams;   // tags are the kept family when cameras are eliminated
    hipLaunchKernelGGL(k_cov_gram, dim3(e.n_tags), dim3(256), 0, e.stream, X, ldb, e.n_pad, e.n_tags, kept,
                       (const double*)e.Le, (const int32_t*)(e.active + e.n_cams), cov_dev);
}

// Touches every kernel of this file once (vmm_ba_create): the code object is loaded and the kernel's resources
// are known before any launch is recorded into a hipGraph (nothing may be loaded lazily under stream capture).
int preload_cov_kernels()
{
    hipFuncAttributes at;
    int bad = 0;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cov_prepare)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cov_identity)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cov_transpose)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_trsm_diag)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_trsm_update)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cov_gram)) != hipSuccess;
    return bad;
}

} // namespace vmm
