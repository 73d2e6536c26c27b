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
// (v_mfma_f64_16x16x4_f64); the (tile, K tile) work units are split evenly over the workgroups ("stream-K")
// and the per-tile partial sums are added in a fixed order.
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

// One thread per observation (E order): M_e = s_e H_e s_e + D_e^2 = L_e L_e^T, z_e = L_e^{-1} s_e g_e and the
// observation's block Z_ef = L_e^{-1} (s_e W_ef s_f) into the dense Z.  Every observation of a pose factors that
// pose's 6x6 block again (91 multiply-adds from L2-resident inputs against the 576 bytes of W and Z the thread
// moves: cheaper than a launch of its own and the dependent kernel boundary behind it); the pose's first
// observation stores L_e, z_e and the rhs column of Z.  A pose without observations owns no thread: nobody reads
// its L_e, and its rows of Z (zero since vmm_ba_create) stay zero -- also what a rank needs for a pose whose
// observations live on another rank.
// SPARSE: the block goes into the compressed Z (Engine::Zc: per eliminated pose a 6 x 6 deg(e) row-major panel at
// 36 * start[e], the pose's observations in E order) instead of into the dense matrix, and the rhs lives in ze only.
template <typename WT, bool SPARSE>
__global__ __launch_bounds__(256) void k_form_z(LmCtl* ctl, int64_t n_obs, int64_t n_pad,
                                                 const int32_t* __restrict__ own,
                                                 const int32_t* __restrict__ other,
                                                 const WT* __restrict__ W0, const WT* __restrict__ W1,
                                                 const double* __restrict__ H_E0, const double* __restrict__ g_E0,
                                                 const int64_t alt_off,
                                                 const double* __restrict__ D2,
                                                 double* __restrict__ Le, double* __restrict__ ze,
                                                 const double* __restrict__ scale, int e_off_pose,
                                                 int f_off_pose, double* __restrict__ Z, int ldz, int zcol,
                                                 const int32_t* __restrict__ e_start)
{
    if (ctl->done)
        return;
    phase_stamp(ctl, 2);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_obs)
        return;
    const WT* __restrict__ W = ctl->w_which ? W1 : W0;   // the buffer that holds J_e^T J_f at x
    const double* __restrict__ H_E = H_E0 + small_sel(ctl, alt_off);
    const double* __restrict__ g_E = g_E0 + small_sel(ctl, alt_off);
    const int e = own[i], f = other[i];
    const double* se = scale + 6 * (int64_t)(e_off_pose + e);
    const double* sf = scale + 6 * (int64_t)(f_off_pose + f);
    double L[36];
    {
        const double* d2 = D2 + 6 * (int64_t)(e_off_pose + e);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 6; ++b)
                L[6 * a + b] = se[a] * H_E[36 * (int64_t)e + 6 * a + b] * se[b];
            L[6 * a + a] += d2[a];
        }
    }
    const bool ok = chol6(L);
    const bool first = i == 0 || own[i - 1] != e;
    if (first) {
        if (!ok)
            ctl->lin_fail = 1;
        double v[6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
            v[a] = se[a] * g_E[6 * (int64_t)e + a];
        // z = L^{-1} v
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double t = v[r];
#pragma unroll
            for (int k = 0; k < r; ++k)
                t -= L[6 * r + k] * v[k];
            v[r] = t / L[6 * r + r];
        }
#pragma unroll
        for (int k = 0; k < 36; ++k)
            Le[36 * (int64_t)e + k] = L[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            ze[6 * (int64_t)e + k] = v[k];
            if (!SPARSE)
                Z[(int64_t)(6 * e + k) * ldz + zcol] = v[k];
        }
    }
    double X[36];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            X[6 * a + b] = se[a] * (double)W[(int64_t)(6 * a + b) * n_pad + i] * sf[b];
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
    int64_t rs = ldz;
    double* zrow = Z + (int64_t)(6 * e) * ldz + 6 * f;
    if (SPARSE) {
        const int es = e_start[e];
        rs = 6 * (int64_t)(e_start[e + 1] - es);
        zrow = Z + 36 * (int64_t)es + 6 * (i - es);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c)
            zrow[(int64_t)r * rs + c] = X[6 * r + c];
}

// ---- symmetric rank-k update on the f64 matrix cores --------------------------------------------
//
// C(I,J) = sum_k Z[k][I]^T Z[k][J] for 128x128 tiles with I >= J.  Workgroup = 4 waves, wave (wi,wj)
// owns a 64x64 quadrant = 4x4 MFMA 16x16 tiles (64 accumulator registers per lane).
// v_mfma_f64_16x16x4_f64 operand maps (one f64 per lane): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15]; result 4 f64 per lane at col = lane&15, row = (lane>>4) + 4*reg
// (cdna_hip_programming.md, "f64 MFMA does NOT use these maps").  Both operands are rows of Z (k-major),
// so a lane reads 16 consecutive doubles of one Z row: the tile is staged k-major in LDS with a row
// stride of 144 doubles so rows k and k+1 of a 32-lane ds_read_b64 group fall into opposite bank halves.
//
// Tile size: Z is far larger than an XCD's L2 and every workgroup streams its own two operand panels,
// so the kernel's HBM traffic is (tiles) x K x (rows + columns of a tile) x 8 B.  With 64x64 tiles that
// was 575 MB per launch at 500x200 (PMC FETCH_SIZE; 5.2 TB/s over the 111 us launch -- bandwidth-bound
// at 8 flop/B); 128x128 tiles double the intensity to 16 flop/B and halve the traffic.
//
// Work decomposition: the unit of work is one K stage (16 rows of Z) of one output tile; every workgroup
// gets a contiguous range of units and its first segment id by blockIdx (host plan, make_syrk_plan):
//   * few tiles (500 x 200: 55 tiles on 512 workgroup slots): "stream-K" -- all units, tile-major, cut into
//     equal ranges, so every workgroup issues the same number of MFMAs whatever the tile count;
//   * more tiles than slots (2000 x 1000: 1128): whole rounds of one tile per workgroup, ordered so that the
//     64 workgroups resident on one XCD (blockIdx % 8) hold 64 consecutive tiles of the row-major list and
//     share their operand panels through that XCD's L2 (PMC FETCH_SIZE: 28.0 -> 12.4 GB per launch), then
//     stream-K for the leftover tiles.
// A workgroup writes one 128x128 partial per (tile) segment of its range; k_reduce_partials sums a tile's
// partials in segment order -- deterministic, no atomics.

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kSyrkKT = kKT;        // 16 rows of Z per LDS stage
constexpr int kSyrkT = kST;         // 128x128 output tiles
constexpr int kSyrkRow = 144;       // LDS row stride (doubles) of a 128-wide stage
constexpr int kSyrkTile = kSyrkT * kSyrkT;

struct SyrkPlanDev {
    int n_tiles, n_kt, n_wg;
    const int32_t* tile_bi;     // [n_tiles] in units of 128 rows
    const int32_t* tile_bj;     // [n_tiles]
    const int64_t* wg_u0;       // [n_wg] first unit of each workgroup (by blockIdx)
    const int64_t* wg_u1;       // [n_wg] one past its last unit
    const int32_t* wg_seg0;     // [n_wg] first segment id of each workgroup
    const int32_t* tile_seg0;   // [n_tiles + 1] first segment id of each tile
    double* partials;           // [n_segments][128*128]
};

__global__ __launch_bounds__(256, 2) void k_syrk_streamk(const LmCtl* ctl, const double* __restrict__ Z, int ldz,
                                                         SyrkPlanDev pl)
{
    if (ctl && ctl->done)
        return;
    __shared__ __attribute__((aligned(16))) double As[2][kSyrkKT * kSyrkRow];
    __shared__ __attribute__((aligned(16))) double Bs[2][kSyrkKT * kSyrkRow];
    const int g = blockIdx.x;
    int64_t u = pl.wg_u0[g];
    const int64_t u_end = pl.wg_u1[g];
    int seg = pl.wg_seg0[g];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int lr = tid >> 5;          // 0..7 (+8 for the second half of a K tile)
    const int lc = (tid & 31) * 4;    // 0..124
    const int fk = lane >> 4, fi = lane & 15;
    while (u < u_end) {
        const int t = (int)(u / pl.n_kt);
        const int kt0 = (int)(u % pl.n_kt);
        const int64_t left = u_end - u;
        const int kt1 = (kt0 + left < pl.n_kt) ? (int)(kt0 + left) : pl.n_kt;
        const int I0 = pl.tile_bi[t] * kSyrkT, J0 = pl.tile_bj[t] * kSyrkT;
        const bool diag = I0 == J0;
        double4_t acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[a][b] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
        double4_t va[2], vb[2];
        auto gload = [&](int kt) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double* zr = Z + (int64_t)(kt * kSyrkKT + lr + 8 * h) * ldz;
                va[h] = *reinterpret_cast<const double4_t*>(zr + I0 + lc);
                vb[h] = *reinterpret_cast<const double4_t*>(zr + J0 + lc);   // diagonal tile: the same lines
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *reinterpret_cast<double4_t*>(&As[buf][(lr + 8 * h) * kSyrkRow + lc]) = va[h];
                *reinterpret_cast<double4_t*>(&Bs[buf][(lr + 8 * h) * kSyrkRow + lc]) = vb[h];
            }
        };
        gload(kt0);
        __syncthreads();   // the previous segment's last stage is fully consumed
        lstore(0);
        __syncthreads();
        // the strictly upper 64x64 quadrant of a diagonal tile is never read: its wave only keeps the barriers
        const bool live = !(diag && wi == 0 && wj == 1);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1)
                gload(kt + 1);   // next stage's loads fly while this stage's 64 MFMAs per wave issue
            const double* Ap = As[buf];
            const double* Bp = Bs[buf];
            if (live) {
#pragma unroll
                for (int ks = 0; ks < kSyrkKT / 4; ++ks) {
                    const int row = (ks * 4 + fk) * kSyrkRow;
                    double a[4], b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a[i] = Ap[row + wi * 64 + 16 * i + fi];
                        b[i] = Bp[row + wj * 64 + 16 * i + fi];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
            if (kt + 1 < kt1)
                lstore(buf ^ 1);
            __syncthreads();
        }
        double* Cb = pl.partials + (size_t)seg * kSyrkTile;
        if (live) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cb[(wi * 64 + a * 16 + fk + 4 * r) * kSyrkT + wj * 64 + b * 16 + fi] = acc[a][b][r];
        }
        u += kt1 - kt0;
        ++seg;
    }
}

struct DiagArgs {        // kept family's damped diagonal blocks and rhs, added in the same pass on one GPU
    const double* H_F;
    const double* g_F;
    const double* scale_F;   // scale + 6 * f_off
    const double* D2_F;      // D2 + 6 * f_off
    int n_red, n_pad;
    int64_t alt_off;         // H_F, g_F live in the copy of the small blocks LmCtl::w_which names
};

// S(tile) = -(sum of the tile's partials, in segment order) [+ damped diagonal blocks / rhs / padding].
// Sixteen workgroups per 128x128 tile: one per 16-row slice of a 64x64 quadrant (the strictly upper quadrant of a
// diagonal tile is skipped).  A thread owns two pairs of neighbouring elements (16-byte loads and stores); 880
// workgroups keep every CU's memory pipe busy (four per tile left the launch at 3.7 TB/s).
template <bool ADD_DIAG>
__global__ __launch_bounds__(256) void k_reduce_partials(const LmCtl* ctl, SyrkPlanDev pl, int ld, int n_rows,
                                                         double* __restrict__ S, DiagArgs da)
{
    if (ctl && ctl->done)
        return;
    const int t = blockIdx.x >> 4;
    const int sub = blockIdx.x & 15;
    const int qi = sub >> 3, qj = (sub >> 2) & 1, rs = sub & 3;
    const int I0 = pl.tile_bi[t] * kSyrkT + 64 * qi, J0 = pl.tile_bj[t] * kSyrkT + 64 * qj;
    if (J0 > I0 || I0 + 16 * rs >= n_rows)
        return;
    const int s0 = pl.tile_seg0[t], s1 = pl.tile_seg0[t + 1];
    const int tr = 16 * rs + (threadIdx.x >> 5), tc = (threadIdx.x & 31) * 2;   // rows tr, tr + 8
    if (ADD_DIAG && ctl) {
        da.H_F += small_sel(ctl, da.alt_off);
        da.g_F += small_sel(ctl, da.alt_off);
    }
    double2 acc[2] = { make_double2(0.0, 0.0), make_double2(0.0, 0.0) };
    const double* p0 = pl.partials + (size_t)(64 * qi + tr) * kSyrkT + 64 * qj + tc;
#pragma unroll 4
    for (int q = s0; q < s1; ++q) {
        const double* pq = p0 + (size_t)q * kSyrkTile;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double2 v = *reinterpret_cast<const double2*>(pq + 8 * i * kSyrkT);
            acc[i].x += v.x;
            acc[i].y += v.y;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = I0 + tr + 8 * i;
        if (row >= n_rows)
            continue;
        double out[2] = { -acc[i].x, -acc[i].y };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = J0 + tc + h;
            double v = out[h];
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
            out[h] = v;
        }
        *reinterpret_cast<double2*>(S + (int64_t)row * ld + J0 + tc) = make_double2(out[0], out[1]);
    }
}

// Adds the kept family's damped diagonal blocks and right-hand side (identical on every rank, so with
// world > 1 it runs after the all-reduce): S_ff += s H_f s + D_f^2, padded diagonal = 1, rhs row += s_f g_f.
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


// ---- block-sparse reduced system: S(f, f') -= sum over the eliminated poses e that see both f and f' --------------
//
// The dense path stores Z with its zero blocks and multiplies them (at 25 % visibility 15/16 of the rank-k flops);
// here Z holds only the blocks of co-observed pairs (Engine::Zc) and the product runs over pairs that share an e --
// what the sparse normal-Cholesky behind ceres::Solve (src/TagReconstructor.cpp:725-738) exploits on projects where an
// image sees a handful of tags (README.md:155-216).  No atomics, fixed summation order:
//   * one workgroup per (kept pose f, column group): its LDS holds the 6 rows of S that belong to f for the group's
//     columns (f' <= f: lower triangle), accumulated over the poses e that see f, in the order of f's observations;
//   * for one e the 36 entries of Z_ef are wave-uniform (scalar registers), a lane owns one column (f', c) of e's
//     panel: six loads, 36 multiply-adds, six read-modify-writes of the LDS accumulator;
//   * column f' belongs to wave f' & 3 of the workgroup for the whole kernel, so no two waves ever touch the same
//     accumulator entry and every entry is summed in e order: bit-repeatable.  A wave finds its columns of e with one
//     ballot over e's neighbour list (any order, any length).
// The group's rows are written once, with the kept family's damped diagonal block and right-hand side (one GPU), and
// with zeros up to the next 64-column boundary: the Cholesky kernels load whole 16x16 tiles of the diagonal blocks.
struct RowArgs {
    const LmCtl* ctl;
    const int32_t* items;       // [n_items][2]: kept pose (n_f = the padding rows), column group
    int group_tags;             // tags per column group (accumulator: 6 x 6 group_tags doubles)
    int n_f;
    const int32_t* f_start;     // [n_f + 1] F-order observation range of every kept pose
    const int32_t* f_other;     // [n_obs]  F order: the eliminated pose
    const int32_t* f2e;         // [n_obs]  F order -> E-order position
    const int32_t* e_start;     // [n_e + 1]
    const int32_t* e_other;     // [n_obs]  E order: the kept pose
    const double* Zc;
    const double* ze;
    double* S;
    int ld;
    DiagArgs da;
    int add_diag;
};

__global__ __launch_bounds__(256, 2) void k_schur_rows(RowArgs a)
{
    if (a.ctl && a.ctl->done)
        return;
    extern __shared__ __attribute__((aligned(16))) double rows_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = a.items[2 * blockIdx.x], g = a.items[2 * blockIdx.x + 1];
    const int n_red = a.da.n_red, n_pad = a.da.n_pad, ld = a.ld;
    if (f >= a.n_f) {
        // padding of the reduced system: rows n_red .. n_pad-1 = unit rows (one GPU; world > 1: zero, k_add_diag sets
        // the ones behind the all-reduce), and the right-hand side's padding entries
        for (int i = n_red + (tid >> 6); i < n_pad; i += 4)
            for (int j = lane; j < n_pad; j += 64)
                a.S[(int64_t)i * ld + j] = (a.add_diag && i == j) ? 1.0 : 0.0;
        for (int j = n_red + tid; j < n_pad; j += 256)
            a.S[(int64_t)n_pad * ld + j] = 0.0;
        return;
    }
    const int lo = g * a.group_tags;
    const int hi = min(lo + a.group_tags, f + 1);   // columns f' in [lo, hi): the lower triangle ends at f
    const int W = 6 * a.group_tags;
    double* acc = rows_smem;                                             // [6][W]
    int* slist = reinterpret_cast<int*>(rows_smem + 6 * W) + w * 128;    // this wave's [64] slots | [64] columns
    for (int i = tid; i < 6 * W; i += 256)
        acc[i] = 0.0;
    __syncthreads();
    const double* __restrict__ Zc = a.Zc;
    double bacc = 0.0;   // wave 0, lanes 0..5 of group 0: (Z^T z)_f
    const int i0 = a.f_start[f], i1 = a.f_start[f + 1];
    for (int idx = i0; idx < i1; ++idx) {
        const int e = __builtin_amdgcn_readfirstlane(a.f_other[idx]);
        const int ie = __builtin_amdgcn_readfirstlane(a.f2e[idx]);
        const int es = __builtin_amdgcn_readfirstlane(a.e_start[e]);
        const int deg = __builtin_amdgcn_readfirstlane(a.e_start[e + 1]) - es;
        const int rs = 6 * deg;
        const double* __restrict__ P = Zc + 36 * (int64_t)es;      // e's panel: 6 rows of 6 deg doubles
        const double* __restrict__ Pf = P + 6 * (ie - es);         // Z_ef: wave-uniform
        double A[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < 6; ++q)
                A[r][q] = Pf[r * rs + q];
        if (g == 0 && w == 0 && lane < 6) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
                t += Pf[r * rs + lane] * a.ze[6 * (int64_t)e + r];
            bacc += t;
        }
        for (int base = 0; base < deg; base += 64) {
            const int slot = base + lane;
            const int fp = slot < deg ? a.e_other[es + slot] : -1;
            const bool own = fp >= lo && fp < hi && (fp & 3) == w;
            const unsigned long long mask = __ballot(own);
            const int cnt = __popcll(mask);
            if (cnt == 0)
                continue;
            if (own) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                slist[rank] = slot;
                slist[64 + rank] = fp - lo;
            }
            __builtin_amdgcn_wave_barrier();   // same wave, LDS in issue order: the list is complete for the reads below
            for (int t0 = 0; t0 < 6 * cnt; t0 += 64) {
                const int t = t0 + lane;
                if (t < 6 * cnt) {
                    const int k = t / 6, c = t - 6 * k;
                    const double* __restrict__ bp = P + 6 * slist[k] + c;
                    double b[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        b[r] = bp[r * rs];
                    double* ap = acc + 6 * slist[64 + k] + c;
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        double o = A[0][q] * b[0];
#pragma unroll
                        for (int r = 1; r < 6; ++r)
                            o = fma(A[r][q], b[r], o);
                        ap[q * W] += o;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the list is overwritten by the next chunk
        }
    }
    __syncthreads();
    if (a.add_diag && a.ctl) {
        a.da.H_F += small_sel(a.ctl, a.da.alt_off);
        a.da.g_F += small_sel(a.ctl, a.da.alt_off);
    }
    // the group's columns of rows 6f .. 6f+5; the last group also zeroes the rest of the diagonal 64-block
    const bool last = hi == f + 1;
    const int c0 = 6 * lo, c1 = last ? min(((6 * f + 6 + 63) / 64) * 64, ld) : 6 * hi;
    const int ncol = c1 - c0;
    for (int i = tid; i < 6 * ncol; i += 256) {
        const int q = i / ncol, col = i - q * ncol;
        const int gcol = c0 + col, row = 6 * f + q;
        double v = (gcol < 6 * hi) ? -acc[q * W + col] : 0.0;
        if (a.add_diag && gcol >= 6 * f && gcol < 6 * f + 6) {
            const int b = gcol - 6 * f;
            v += a.da.scale_F[row] * a.da.H_F[36 * (int64_t)f + 6 * q + b] * a.da.scale_F[gcol];
            if (q == b)
                v += a.da.D2_F[row];
        }
        a.S[(int64_t)row * ld + gcol] = v;
    }
    if (g == 0 && w == 0 && lane < 6) {
        const int col = 6 * f + lane;
        a.S[(int64_t)n_pad * ld + col] = (a.add_diag ? a.da.scale_F[col] * a.da.g_F[col] : 0.0) - bacc;
    }
}

// ---- launchers -----------------------------------------------------------------------------------

static SyrkPlanDev plan_dev(const SyrkPlan& p)
{
    SyrkPlanDev d;
    d.n_tiles = p.n_tiles;
    d.n_kt = p.n_kt;
    d.n_wg = p.n_wg;
    d.tile_bi = p.tile_bi;
    d.tile_bj = p.tile_bj;
    d.wg_u0 = p.wg_u0;
    d.wg_u1 = p.wg_u1;
    d.wg_seg0 = p.wg_seg0;
    d.tile_seg0 = p.tile_seg0;
    d.partials = p.partials;
    return d;
}

void launch_elim(Engine& e)
{
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const double* H_E = e.elim_cams ? e.H_cam : e.H_tag;
    const double* g_E = e.elim_cams ? e.g_cam : e.g_tag;
    if (e.ordE.n > 0) {
        const dim3 grid((unsigned)((e.ordE.n + 255) / 256));
        double* const Zout = e.sparse_schur ? e.Zc : e.Z;
#define VMM_FORM_Z(WT, SP, W0, W1)                                                                                      \
    hipLaunchKernelGGL((k_form_z<WT, SP>), grid, dim3(256), 0, e.stream, e.ctl, e.ordE.n, e.ordE.n_pad, e.ordE.own,      \
                       e.ordE.other, (const WT*)(W0), (const WT*)(W1), H_E, g_E, e.small_alt_off, (const double*)e.D2,   \
                       e.Le, e.ze, e.scale, e_off, f_off, Zout, e.ldz, e.n_pad, (const int32_t*)e.ordE.start)
        if (e.f32_accum) {
            if (e.sparse_schur)
                VMM_FORM_Z(float, true, e.Wf, e.Wf2);
            else
                VMM_FORM_Z(float, false, e.Wf, e.Wf2);
        } else {
            if (e.sparse_schur)
                VMM_FORM_Z(double, true, e.W, e.W2);
            else
                VMM_FORM_Z(double, false, e.W, e.W2);
        }
#undef VMM_FORM_Z
    }
}

void launch_syrk_plan(hipStream_t st, const LmCtl* ctl, const double* Z, int ldz, const SyrkPlan& p)
{
    if (p.n_wg > 0)
        hipLaunchKernelGGL(k_syrk_streamk, dim3(p.n_wg), dim3(256), 0, st, ctl, Z, ldz, plan_dev(p));
}

void launch_reduce_plan(hipStream_t st, const LmCtl* ctl, const SyrkPlan& p, int ld, int n_rows, double* S)
{
    DiagArgs da = {};
    if (p.n_tiles > 0)
        hipLaunchKernelGGL((k_reduce_partials<false>), dim3(16 * p.n_tiles), dim3(256), 0, st, ctl, plan_dev(p), ld, n_rows,
                           S, da);
}

void launch_schur_rows(Engine& e, bool add_diag)
{
    if (e.n_row_items <= 0)
        return;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    RowArgs a;
    a.ctl = e.ctl;
    a.items = e.row_items;
    a.group_tags = e.row_group_tags;
    a.n_f = e.n_f;
    a.f_start = e.ordF.start;
    a.f_other = e.ordF.other;
    a.f2e = e.f2e;
    a.e_start = e.ordE.start;
    a.e_other = e.ordE.other;
    a.Zc = e.Zc;
    a.ze = e.ze;
    a.S = e.S;
    a.ld = e.ldz;
    a.da.H_F = e.elim_cams ? e.H_tag : e.H_cam;
    a.da.g_F = e.elim_cams ? e.g_tag : e.g_cam;
    a.da.scale_F = e.scale + 6 * (size_t)f_off;
    a.da.D2_F = e.D2 + 6 * (size_t)f_off;
    a.da.n_red = e.n_red;
    a.da.n_pad = e.n_pad;
    a.da.alt_off = e.small_alt_off;
    a.add_diag = add_diag ? 1 : 0;
    const size_t lds = sizeof(double) * 36 * (size_t)e.row_group_tags + sizeof(int) * 4 * 128;
    hipLaunchKernelGGL(k_schur_rows, dim3(e.n_row_items), dim3(256), lds, e.stream, a);
}

void launch_syrk_only(Engine& e)
{
    if (e.sparse_schur)
        launch_schur_rows(e, !e.multi);
    else
        launch_syrk_plan(e.stream, e.ctl, e.Z, e.ldz, e.syrk);
}

// S = -(sum of partials) [+ damped diagonal blocks and rhs on one GPU; with world > 1 they are added by
// launch_add_diag after the all-reduce, being identical on every rank]
void launch_syrk_reduced(Engine& e)
{
    launch_syrk_only(e);
    if (e.sparse_schur)
        return;   // k_schur_rows writes S itself (with the diagonal blocks on one GPU)
    if (e.multi) {
        launch_reduce_plan(e.stream, e.ctl, e.syrk, e.ldz, e.n_pad + 1, e.S);
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
    da.alt_off = e.small_alt_off;
    hipLaunchKernelGGL((k_reduce_partials<true>), dim3(16 * e.syrk.n_tiles), dim3(256), 0, e.stream, e.ctl,
                       plan_dev(e.syrk), e.ldz, e.n_pad + 1, e.S, da);
}

// World > 1: only the lower triangle of the reduced system (rows 0..n_pad, row r holds r+1 entries, the rhs row
// n_pad holds n_pad+1) travels through the all-reduce: (n_pad+1)(n_pad+2)/2 doubles instead of (n_pad+1) * ld.
__global__ __launch_bounds__(256) void k_pack_lower(const LmCtl* ctl, const double* __restrict__ S, int ld, int n_rows,
                                                    double* __restrict__ packed, bool unpack, double* __restrict__ Sout)
{
    if (ctl && ctl->done)
        return;
    const int r = blockIdx.x;
    if (r >= n_rows)
        return;
    const int64_t base = (int64_t)r * (r + 1) / 2;
    for (int c = threadIdx.x; c <= r; c += 256) {
        if (unpack)
            Sout[(int64_t)r * ld + c] = packed[base + c];
        else
            packed[base + c] = S[(int64_t)r * ld + c];
    }
}

void launch_pack_lower(Engine& e, bool unpack)
{
    if (!e.multi)
        return;
    hipLaunchKernelGGL(k_pack_lower, dim3(e.n_pad + 1), dim3(256), 0, e.stream, (const LmCtl*)e.ctl, (const double*)e.S,
                       e.ldz, e.n_pad + 1, e.S_packed, unpack, e.S);
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

// Touches every kernel of this file once (vmm_ba_create): the code object is loaded and the kernel's resources
// are known before any launch is recorded into a hipGraph (nothing may be loaded lazily under stream capture).
int preload_schur_kernels()
{
    hipFuncAttributes at;
    int bad = 0;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_form_z<double, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_form_z<float, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_form_z<double, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_form_z<float, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_schur_rows)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_syrk_streamk)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_reduce_partials<true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_reduce_partials<false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_add_diag)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_pack_lower)) != hipSuccess;
    return bad;
}

} // namespace vmm
