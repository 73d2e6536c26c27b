// Dense Cholesky factorisation + triangular solves of the reduced system (gfx950).
//
// Replaces the numeric Cholesky of the linear solver Ceres runs inside ceres::Solve for the reference
// (src/TagReconstructor.cpp:737-738; SPARSE_NORMAL_CHOLESKY / DENSE_QR by default, both exact).
//
// Matrix layout: row-major, leading dimension ld, only the lower block triangle is used.  Order
// n_pad = 64 * n_blk; the right-hand side is stored as ROW n_pad of the same array, so the blocked
// right-looking factorisation also performs the forward substitution (row n_pad ends as (L^-1 b)^T).
// Per block column k: (1) panel kernel -- every workgroup re-factors the 64x64 diagonal block in LDS
// (cheaper than a dependent launch), then solves X L_kk^T = A_ik for 256 rows per workgroup with one
// thread per row, and also stores the panel transposed (P, 64 x ld) so that (2) the trailing update
// A_ij -= L_ik L_jk^T is the same k-major MFMA f64 rank-k kernel that forms the Schur complement.
// The back-substitution L^T y = w runs one small kernel per block, last block first.
#include "engine.hpp"

namespace vmm {

void launch_syrk_raw(hipStream_t st, const LmCtl* ctl, const double* Z, int ldz, int row_blk0, int n_row_blk,
                     int col_blk0, int n_col_blk, int split_k, int k_chunk, double* C, int ldc,
                     size_t slab_stride, bool subtract);

constexpr int kLd = 65;  // LDS row stride of the 64x64 diagonal block (odd -> conflict-free columns)

// Cholesky of the 64x64 block held in LDS (lower triangle, stride kLd), 256 threads.
// invd[j] = 1 / L[j][j].  Returns false when a pivot is not positive.
__device__ __forceinline__ bool potrf64_lds(double* A, double* invd)
{
    const int tid = threadIdx.x;
    const int i = tid & 63, grp = tid >> 6;
    bool ok = true;
    for (int j = 0; j < 64; ++j) {
        const double ajj = A[j * kLd + j];
        const bool good = (ajj > 0.0) && isfinite(ajj);
        ok = ok && good;
        const double piv = good ? ajj : 1.0;
        const double d = sqrt(piv);
        const double inv = 1.0 / d;
        double lij = 0.0;
        if (i > j)
            lij = A[i * kLd + j] * inv;
        __syncthreads();  // everyone has read column j before it is overwritten
        if (grp == 0) {
            if (i > j)
                A[i * kLd + j] = lij;
            else if (i == j) {
                A[j * kLd + j] = d;
                invd[j] = inv;
            }
        }
        __syncthreads();
        // trailing update: A[i][c] -= L[i][j] L[c][j] for j < c <= i; wave = column group
        if (i > j) {
            for (int c = j + 1 + grp; c <= i; c += 4)
                A[i * kLd + c] -= lij * A[c * kLd + j];
        }
        // no barrier needed here: the next iteration's reads of column j+1 happen after the
        // barrier that follows them only for writes; keep a barrier for the RAW on A[.][j+1]
        __syncthreads();
    }
    return ok;
}

__global__ __launch_bounds__(256) void k_chol_panel(LmCtl* ctl, double* __restrict__ S, int ld, int n_pad,
                                                    int k, double* __restrict__ P)
{
    if (ctl->done || ctl->lin_fail)
        return;
    __shared__ double A[64 * kLd];
    __shared__ double invd[64];
    const int tid = threadIdx.x;
    const int K0 = k * kNB;
    // load the lower triangle of the diagonal block
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        A[r * kLd + c] = (c <= r) ? S[(int64_t)(K0 + r) * ld + K0 + c] : 0.0;
    }
    __syncthreads();
    const bool ok = potrf64_lds(A, invd);
    if (blockIdx.x == 0) {
        if (!ok && tid == 0)
            ctl->lin_fail = 1;
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            if (c <= r)
                S[(int64_t)(K0 + r) * ld + K0 + c] = A[r * kLd + c];
        }
        return;
    }
    // rows below the diagonal block, including the rhs row n_pad
    const int row = K0 + kNB + (blockIdx.x - 1) * 256 + tid;
    if (row > n_pad)
        return;
    double x[64];
    double* srow = S + (int64_t)row * ld + K0;
#pragma unroll
    for (int c = 0; c < 64; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(srow + c);
        x[c] = v.x;
        x[c + 1] = v.y;
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        double s = x[j];
#pragma unroll
        for (int m = 0; m < j; ++m)
            s -= x[m] * A[j * kLd + m];
        x[j] = s * invd[j];
    }
#pragma unroll
    for (int c = 0; c < 64; c += 2)
        *reinterpret_cast<double2*>(srow + c) = make_double2(x[c], x[c + 1]);
#pragma unroll
    for (int c = 0; c < 64; ++c)
        P[(int64_t)c * ld + row] = x[c];
}

// One step of L^T y = w (w lives in row n_pad of S).  Launched for kb = n_blk-1 .. 0 with kb+1
// workgroups of 64 threads: workgroup m first applies y_{kb+1} to w_m, then workgroup kb solves its
// diagonal block.
__global__ __launch_bounds__(64) void k_backsolve_step(const LmCtl* ctl, double* __restrict__ S, int ld,
                                                       int n_pad, int n_blk, int kb, double* __restrict__ y)
{
    if (ctl->done || ctl->lin_fail)
        return;
    const int m = blockIdx.x;
    const int c = threadIdx.x;
    double* w = S + (int64_t)n_pad * ld;
    double wc = w[m * kNB + c];
    if (kb + 1 < n_blk) {
        const int R0 = (kb + 1) * kNB;
        const double* Lb = S + (int64_t)R0 * ld + m * kNB + c;
        double acc = 0.0;
#pragma unroll 8
        for (int r = 0; r < 64; ++r)
            acc += Lb[(int64_t)r * ld] * y[R0 + r];
        wc -= acc;
        w[m * kNB + c] = wc;
    }
    if (m != kb)
        return;
    __shared__ double L[64 * kLd];
    const int K0 = kb * kNB;
    for (int r = 0; r < 64; ++r)
        L[r * kLd + c] = (c <= r) ? S[(int64_t)(K0 + r) * ld + K0 + c] : 0.0;
    __syncthreads();
    double yj = 0.0;
    for (int j = 63; j >= 0; --j) {
        const double wj = __shfl(wc, j, 64);
        const double v = wj / L[j * kLd + j];
        if (c == j)
            yj = v;
        if (c < j)
            wc -= L[j * kLd + c] * v;
    }
    y[K0 + c] = yj;
}

void launch_cholesky_solve(Engine& e, double* S, int n_pad, int ld, double* y, LmCtl* ctl)
{
    const int n_blk = n_pad / kNB;
    for (int k = 0; k < n_blk; ++k) {
        const int rows_below = n_pad + 1 - (k + 1) * kNB;
        const int wgs = 1 + (rows_below + 255) / 256;
        hipLaunchKernelGGL(k_chol_panel, dim3(wgs), dim3(256), 0, e.stream, ctl, S, ld, n_pad, k, e.P);
        // trailing update on block rows k+1..n_blk (rhs row block included), block cols k+1..n_blk-1
        if (k + 1 < n_blk)
            launch_syrk_raw(e.stream, ctl, e.P, ld, k + 1, n_blk - k, k + 1, n_blk - 1 - k, 1, kNB, S, ld, 0,
                            true);
    }
    for (int kb = n_blk - 1; kb >= 0; --kb)
        hipLaunchKernelGGL(k_backsolve_step, dim3(kb + 1), dim3(64), 0, e.stream, ctl, S, ld, n_pad, n_blk, kb, y);
}

} // namespace vmm
