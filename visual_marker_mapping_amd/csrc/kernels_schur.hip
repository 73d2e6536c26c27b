// Block elimination of one pose family and formation of the dense reduced system (gfx950).
//
// Replaces the numeric phase of the exact sparse normal-equation solve Ceres runs for the reference
// (ceres::Solve at src/TagReconstructor.cpp:737-738; elimination ordering :675-676,695-696): with
// e-blocks = poses of the eliminated family and f-blocks = poses of the kept family,
//     M_e = s_e H_e s_e + D_e^2 = L_e L_e^T,   Z_ef = L_e^{-1} (s_e W_ef s_f),   z_e = L_e^{-1} s_e g_e
//     S   = diag_f(s H_f s + D_f^2) - Z^T Z,   b = s_f g_f - Z^T z
// (s = Jacobi column scaling, D^2 = LM diagonal / radius).
// Z is stored dense, row-major [6 n_e (+pad)] x ldz with the rhs z as one extra column, so that
// S and b come out of ONE symmetric rank-k update computed on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64), split along K into slabs that are summed in a fixed order.
#include "engine.hpp"

namespace vmm {

// ---- E-block factorisation ---------------------------------------------------------------------

__device__ __forceinline__ bool chol6(double* A)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
#pragma unroll
        for (int k = 0; k < j; ++k)
            d -= A[6 * j + k] * A[6 * j + k];
        if (!(d > 0.0) || !isfinite(d)) {
            ok = false;
            d = 1.0;
        }
        d = sqrt(d);
        A[6 * j + j] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; ++k)
                s -= A[6 * i + k] * A[6 * j + k];
            A[6 * i + j] = s * inv;
        }
    }
    return ok;
}

// One thread per eliminated pose.
__global__ void k_elim_factor(LmCtl* ctl, int n_e, int e_off_pose, const double* __restrict__ H_E,
                              const double* __restrict__ g_E, const double* __restrict__ scale,
                              const double* __restrict__ D2, double* __restrict__ Le,
                              double* __restrict__ ze, const int32_t* __restrict__ pose_task,
                              double* __restrict__ Z, int ldz, int zcol)
{
    if (ctl->done)
        return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_e)
        return;
    const double* s = scale + 6 * (int64_t)(e_off_pose + e);
    const double* d2 = D2 + 6 * (int64_t)(e_off_pose + e);
    double M[36], v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b < 6; ++b)
            M[6 * a + b] = s[a] * H_E[36 * (int64_t)e + 6 * a + b] * s[b];
        M[6 * a + a] += d2[a];
        v[a] = s[a] * g_E[6 * (int64_t)e + a];
    }
    if (!chol6(M))
        ctl->lin_fail = 1;
    // z = L^{-1} v
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = v[i];
#pragma unroll
        for (int k = 0; k < i; ++k)
            t -= M[6 * i + k] * v[k];
        v[i] = t / M[6 * i + i];
    }
#pragma unroll
    for (int k = 0; k < 36; ++k)
        Le[36 * (int64_t)e + k] = M[k];
    // z column of the augmented Z (rhs of the eliminated block).  A pose is owned by the rank that
    // holds its observations; elsewhere its rows of Z stay zero.
    const bool owned = pose_task[e + 1] > pose_task[e];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ze[6 * (int64_t)e + k] = v[k];
        Z[(int64_t)(6 * e + k) * ldz + zcol] = owned ? v[k] : 0.0;
    }
}

// One thread per observation (E order): Z block = L_e^{-1} (s_e W s_f) into the dense Z.
__global__ __launch_bounds__(256) void k_form_z(const LmCtl* ctl, int64_t n_obs, int64_t n_pad,
                                                 const int32_t* __restrict__ own,
                                                 const int32_t* __restrict__ other,
                                                 const double* __restrict__ W,
                                                 const double* __restrict__ Le,
                                                 const double* __restrict__ scale, int e_off_pose,
                                                 int f_off_pose, double* __restrict__ Z, int ldz)
{
    if (ctl->done)
        return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_obs)
        return;
    const int e = own[i], f = other[i];
    const double* se = scale + 6 * (int64_t)(e_off_pose + e);
    const double* sf = scale + 6 * (int64_t)(f_off_pose + f);
    const double* L = Le + 36 * (int64_t)e;
    double X[36];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            X[6 * a + b] = se[a] * W[(int64_t)(6 * a + b) * n_pad + i] * sf[b];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double inv = 1.0 / L[6 * r + r];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double t = X[6 * r + c];
#pragma unroll
            for (int k = 0; k < r; ++k)
                t -= L[6 * r + k] * X[6 * k + c];
            X[6 * r + c] = t * inv;
        }
    }
    double* zrow = Z + (int64_t)(6 * e) * ldz + 6 * f;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c)
            zrow[(int64_t)r * ldz + c] = X[6 * r + c];
}

// ---- symmetric rank-k update on the f64 matrix cores --------------------------------------------
//
// C(I,J) (+)= sum_k Z[k][I]^T Z[k][J] for 64x64 tiles with I >= J.  Workgroup = 4 waves, wave (wi,wj)
// owns a 32x32 quadrant = 2x2 MFMA 16x16 tiles.  v_mfma_f64_16x16x4_f64 operand maps (one f64 per
// lane): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; result 4 f64 per lane at
// col = lane&15, row = (lane>>4) + 4*reg (cdna_hip_programming.md, "f64 MFMA does NOT use these maps").
// Both operands are rows of Z (k-major), so a lane reads 16 consecutive doubles of one Z row: the
// tile is staged k-major in LDS with a row stride of 80 doubles so rows k and k+1 of a 32-lane
// ds_read_b64 group fall into opposite bank halves.

typedef double double4_t __attribute__((ext_vector_type(4)));

enum { SYRK_SLAB = 0, SYRK_SUB = 1 };

template <int MODE>
__global__ __launch_bounds__(256) void k_syrk(const LmCtl* ctl, const double* __restrict__ Z, int ldz,
                                              int row_blk0, int n_row_blk, int col_blk0, int n_col_blk,
                                              int k_chunk, double* __restrict__ C, int ldc,
                                              size_t slab_stride)
{
    if (ctl && ctl->done)
        return;
    // blockIdx.x enumerates lower-triangular tiles (bi >= bj) of the requested block range,
    // blockIdx.y the K split.
    int t = blockIdx.x;
    int bi = 0, bj = 0;
    {
        // tile list: for each row block r (absolute index), columns col_blk0 .. min(r, col_end-1)
        const int col_end = col_blk0 + n_col_blk;
        bool found = false;
        for (int r = row_blk0; r < row_blk0 + n_row_blk; ++r) {
            const int last = (r < col_end - 1) ? r : col_end - 1;
            const int cnt = last - col_blk0 + 1;
            if (cnt <= 0)
                continue;
            if (t < cnt) {
                bi = r;
                bj = col_blk0 + t;
                found = true;
                break;
            }
            t -= cnt;
        }
        if (!found)
            return;  // uniform per workgroup: the host sizes the grid to the exact tile count
    }
    const int k0 = blockIdx.y * k_chunk;
    const int I0 = bi * kNB, J0 = bj * kNB;
    const bool diag = (bi == bj);

    // K tiles of 32 rows, double-buffered in LDS; the next tile's global loads are issued before the
    // current tile's 32 MFMAs per wave so that the memory latency hides behind the wave's own compute.
    constexpr int KT = 32;
    __shared__ __attribute__((aligned(16))) double As[2][KT * kLdsRow];
    __shared__ __attribute__((aligned(16))) double Bs[2][KT * kLdsRow];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    // global->LDS: 32 rows x 64 cols = 2048 doubles per operand, 256 threads x 2 x double4
    const int lr = tid >> 4;          // 0..15 (+16 for the second half)
    const int lc = (tid & 15) * 4;    // 0..60
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            acc[a][b] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };

    const int fk = lane >> 4, fi = lane & 15;
    const int n_kt = k_chunk / KT;      // host guarantees k_chunk % 32 == 0
    double4_t va[2], vb[2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double* zr = Z + (int64_t)(k0 + kt * KT + lr + 16 * h) * ldz;
            va[h] = *reinterpret_cast<const double4_t*>(zr + I0 + lc);
            vb[h] = diag ? va[h] : *reinterpret_cast<const double4_t*>(zr + J0 + lc);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<double4_t*>(&As[buf][(lr + 16 * h) * kLdsRow + lc]) = va[h];
            if (!diag)
                *reinterpret_cast<double4_t*>(&Bs[buf][(lr + 16 * h) * kLdsRow + lc]) = vb[h];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < n_kt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_kt)
            gload(kt + 1);
        const double* Ap = As[buf];
        const double* Bp = diag ? As[buf] : Bs[buf];
#pragma unroll
        for (int ks = 0; ks < KT / 4; ++ks) {
            const int row = (ks * 4 + fk) * kLdsRow;
            const double a0 = Ap[row + wi * 32 + fi];
            const double a1 = Ap[row + wi * 32 + 16 + fi];
            const double b0 = Bp[row + wj * 32 + fi];
            const double b1 = Bp[row + wj * 32 + 16 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < n_kt)
            lstore(buf ^ 1);
        __syncthreads();
    }
    double* Cb = C + (MODE == SYRK_SLAB ? (size_t)blockIdx.y * slab_stride : 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I0 + wi * 32 + a * 16 + fk + 4 * r;
                const int col = J0 + wj * 32 + b * 16 + fi;
                double* p = Cb + (int64_t)row * ldc + col;
                if (MODE == SYRK_SLAB)
                    *p = acc[a][b][r];
                else
                    *p -= acc[a][b][r];
            }
}

// number of lower tiles with row block in [row_blk0, row_blk0+n_row_blk) and column block in
// [col_blk0, col_blk0+n_col_blk), bj <= bi
static int count_tiles(int row_blk0, int n_row_blk, int col_blk0, int n_col_blk)
{
    int n = 0;
    for (int r = row_blk0; r < row_blk0 + n_row_blk; ++r) {
        const int last = (r < col_blk0 + n_col_blk - 1) ? r : col_blk0 + n_col_blk - 1;
        if (last >= col_blk0)
            n += last - col_blk0 + 1;
    }
    return n;
}

void launch_syrk_raw(hipStream_t st, const LmCtl* ctl, const double* Z, int ldz, int row_blk0, int n_row_blk,
                     int col_blk0, int n_col_blk, int split_k, int k_chunk, double* C, int ldc,
                     size_t slab_stride, bool subtract)
{
    const int tiles = count_tiles(row_blk0, n_row_blk, col_blk0, n_col_blk);
    if (tiles <= 0)
        return;
    if (subtract)
        hipLaunchKernelGGL((k_syrk<SYRK_SUB>), dim3(tiles, 1), dim3(256), 0, st, ctl, Z, ldz, row_blk0,
                           n_row_blk, col_blk0, n_col_blk, k_chunk, C, ldc, slab_stride);
    else
        hipLaunchKernelGGL((k_syrk<SYRK_SLAB>), dim3(tiles, split_k), dim3(256), 0, st, ctl, Z, ldz, row_blk0,
                           n_row_blk, col_blk0, n_col_blk, k_chunk, C, ldc, slab_stride);
}

// S = -(sum of slabs) over the lower block triangle (row blocks 0..n_blk incl. the rhs row block).
struct DiagArgs {        // kept family's damped diagonal blocks and rhs, added in the same pass on one GPU
    const double* H_F;
    const double* g_F;
    const double* scale_F;   // scale + 6 * f_off
    const double* D2_F;      // D2 + 6 * f_off
    int n_red, n_pad;
};

template <bool ADD_DIAG>
__global__ __launch_bounds__(256) void k_reduce_slabs(const LmCtl* ctl, const double* __restrict__ slabs,
                                                      int split_k, size_t slab_stride, int ld, int n_rows,
                                                      int n_cols, double* __restrict__ S, DiagArgs da)
{
    if (ctl && ctl->done)
        return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(idx / n_cols), col = (int)(idx % n_cols);
    if (row >= n_rows)
        return;
    if ((col / kNB) > (row / kNB))
        return;
    const int64_t off = (int64_t)row * ld + col;
    double s = 0.0;
    for (int k = 0; k < split_k; ++k)
        s += slabs[(size_t)k * slab_stride + off];
    double v = -s;
    if (ADD_DIAG) {
        if (row == da.n_pad) {                       // rhs row: b = s_f g_f - Z^T z
            if (col < da.n_red)
                v += da.scale_F[col] * da.g_F[col];
        } else if (row < da.n_red && col <= row && (row / 6) == (col / 6)) {
            const int f = row / 6, a = row % 6, b = col % 6;
            v += da.scale_F[row] * da.H_F[36 * (int64_t)f + 6 * a + b] * da.scale_F[col];
            if (a == b)
                v += da.D2_F[row];
        } else if (row >= da.n_red && row < da.n_pad && col == row) {
            v = 1.0;                                 // padding of the reduced system
        }
    }
    S[off] = v;
}

// Adds the kept family's damped diagonal blocks and right-hand side (identical on every rank, so it
// runs after the all-reduce): S_ff += s H_f s + D_f^2, padded diagonal = 1, rhs row += s_f g_f.
__global__ void k_add_diag(const LmCtl* ctl, int n_f, int f_off_pose, const double* __restrict__ H_F,
                           const double* __restrict__ g_F, const double* __restrict__ scale,
                           const double* __restrict__ D2, double* __restrict__ S, int ld, int n_red,
                           int n_pad)
{
    if (ctl && ctl->done)
        return;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < 36 * n_f) {
        const int f = tid / 36, a = (tid % 36) / 6, b = tid % 6;
        if (b <= a) {
            const double* s = scale + 6 * (int64_t)(f_off_pose + f);
            double v = s[a] * H_F[36 * (int64_t)f + 6 * a + b] * s[b];
            if (a == b)
                v += D2[6 * (int64_t)(f_off_pose + f) + a];
            S[(int64_t)(6 * f + a) * ld + 6 * f + b] += v;
        }
        if (b == 0)
            S[(int64_t)n_pad * ld + 6 * f + a] += scale[6 * (int64_t)(f_off_pose + f) + a] * g_F[6 * (int64_t)f + a];
    } else {
        const int i = n_red + (tid - 36 * n_f);
        if (i < n_pad)
            S[(int64_t)i * ld + i] = 1.0;
    }
}

// ---- launchers -----------------------------------------------------------------------------------

void launch_elim(Engine& e)
{
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const double* H_E = e.elim_cams ? e.H_cam : e.H_tag;
    const double* g_E = e.elim_cams ? e.g_cam : e.g_tag;
    hipLaunchKernelGGL(k_elim_factor, dim3((e.n_e + 63) / 64), dim3(64), 0, e.stream, e.ctl, e.n_e, e_off, H_E, g_E,
                       e.scale, e.D2, e.Le, e.ze, e.ordE.pose_task, e.Z, e.ldz, e.n_pad);
    if (e.ordE.n > 0)
        hipLaunchKernelGGL(k_form_z, dim3((unsigned)((e.ordE.n + 255) / 256)), dim3(256), 0, e.stream, e.ctl,
                           e.ordE.n, e.ordE.n_pad, e.ordE.own, e.ordE.other, e.W, e.Le, e.scale, e_off, f_off,
                           e.Z, e.ldz);
}

void launch_reduce_slabs(hipStream_t st, const LmCtl* ctl, const double* slabs, int split_k, size_t slab_stride,
                         int ld, int n_rows, int n_cols, double* S)
{
    const int64_t total = (int64_t)n_rows * n_cols;
    DiagArgs da = {};
    hipLaunchKernelGGL((k_reduce_slabs<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctl, slabs,
                       split_k, slab_stride, ld, n_rows, n_cols, S, da);
}

// slabs[s] = (Z^T Z)(K chunk s): row blocks 0..n_blk (the last one holds the rhs row), column blocks 0..n_blk-1
void launch_syrk_only(Engine& e)
{
    launch_syrk_raw(e.stream, e.ctl, e.Z, e.ldz, 0, e.n_blk + 1, 0, e.n_blk, e.split_k, e.k_chunk, e.slabs,
                    e.ldz, (size_t)e.ldz * e.ldz, false);
}

// S = -(sum of slabs) [+ damped diagonal blocks and rhs on one GPU; with world > 1 they are added by
// launch_add_diag after the all-reduce, being identical on every rank]
void launch_syrk_reduced(Engine& e)
{
    launch_syrk_only(e);
    if (e.multi) {
        launch_reduce_slabs(e.stream, e.ctl, e.slabs, e.split_k, (size_t)e.ldz * e.ldz, e.ldz, e.n_pad + 1, e.n_pad,
                            e.S);
        return;
    }
    const int f_off = e.elim_cams ? e.n_cams : 0;
    DiagArgs da;
    da.H_F = e.elim_cams ? e.H_tag : e.H_cam;
    da.g_F = e.elim_cams ? e.g_tag : e.g_cam;
    da.scale_F = e.scale + 6 * (size_t)f_off;
    da.D2_F = e.D2 + 6 * (size_t)f_off;
    da.n_red = e.n_red;
    da.n_pad = e.n_pad;
    const int64_t total = (int64_t)(e.n_pad + 1) * e.n_pad;
    hipLaunchKernelGGL((k_reduce_slabs<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e.stream, e.ctl,
                       e.slabs, e.split_k, (size_t)e.ldz * e.ldz, e.ldz, e.n_pad + 1, e.n_pad, e.S, da);
}

void launch_add_diag(Engine& e)
{
    if (!e.multi)
        return;   // folded into the slab reduction
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const double* H_F = e.elim_cams ? e.H_tag : e.H_cam;
    const double* g_F = e.elim_cams ? e.g_tag : e.g_cam;
    const int threads = 36 * e.n_f + (e.n_pad - e.n_red);
    hipLaunchKernelGGL(k_add_diag, dim3((threads + 255) / 256), dim3(256), 0, e.stream, e.ctl, e.n_f, f_off, H_F, g_F,
                       e.scale, e.D2, e.S, e.ldz, e.n_red, e.n_pad);
}

} // namespace vmm
