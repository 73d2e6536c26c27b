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
// SPARSE: the block goes into the compressed Z (Engine::Zc: one COLUMN-major 6x6 block per observation, E order: a
// column's six doubles are contiguous, so a lane of k_schur_pairs fetches its right-operand column with three 16-byte
// loads) instead of into the dense matrix, and the rhs lives in ze only.
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
        double* zb = Z + 36 * i;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int r = 0; r < 6; ++r)
                zb[6 * c + r] = X[6 * r + c];
        return;
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

// ---- the same product with ONE 8-wave workgroup per CU (few tiles: 500 x 200 has 55 on 256 CUs) -------------------
//
// k_syrk_streamk fills the chip with two 4-wave workgroups per CU and nine K slices per tile: 495 partial tiles of
// 128 KB, written in one burst at the end (13 us of a 104 us launch) and read back by k_reduce_partials.  Here the two
// workgroups of a CU are one: waves 0-3 multiply the even 16-row stages of the workgroup's K slice, waves 4-7 the odd ones
// (each half with its own stage buffers, half a stage apart: syrk_wide_item), and the two halves add their accumulators
// through the LDS before the one partial tile of the workgroup is stored -- half the partial tiles (3-5 per output tile),
// half the store burst, half of what the reduction reads.
// A diagonal tile needs only the 36 MFMA tiles on and below its diagonal: wave w of a half takes tile rows w and 7 - w of
// the 8 x 8 grid (w + 1 and 8 - w tiles: nine each), instead of one 64 x 64 quadrant each with the upper one dead and the
// lower one setting the pace; the host plan gives a diagonal tile 9/16 of the workgroups of an off-diagonal one.
// The strictly upper MFMA tiles inside the two diagonal 64 x 64 blocks are written as mirrors of the lower ones, so the
// reduction finds the full diagonal blocks it found before.
constexpr int kWideRows = 2 * kSyrkKT;
constexpr int kWideStage = kWideRows * kSyrkRow;   // doubles per operand stage

template <bool DIAG, int W>
struct WideTiles {
    static constexpr int NT = DIAG ? 9 : 16, NA = DIAG ? 2 : 4, NB = DIAG ? 8 - W : 4;
    static __device__ __host__ constexpr int arow(int i) { return DIAG ? (i == 0 ? W : 7 - W) : 4 * (W >> 1) + i; }
    static __device__ __host__ constexpr int bcol(int j) { return DIAG ? j : 4 * (W & 1) + j; }
    static __device__ __host__ constexpr int ai(int n) { return DIAG ? (n <= W ? 0 : 1) : (n >> 2); }
    static __device__ __host__ constexpr int bi(int n) { return DIAG ? (n <= W ? n : n - (W + 1)) : (n & 3); }
};

template <bool DIAG, int W>
__device__ __forceinline__ void syrk_wide_item(const double* __restrict__ Z, const int ldz, const int I0, const int J0,
                                               const int kt0, const int kt1, double* smem, double* __restrict__ Cb)
{
    using T = WideTiles<DIAG, W>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = tid >> 8;             // half 0: the even 16-row stages of the item's K range, half 1: the odd ones
    const int t = tid & 255;
    const int lr = t >> 5;              // 0..7 (+8 for the second eight rows of a stage)
    const int lc = (t & 31) * 4;        // 0..124
    const int fk = lane >> 4, fi = lane & 15;
    // each half has its own pair of stage buffers, [buf][A | B][16][kSyrkRow], and loads its own stages
    constexpr int kGStage = kSyrkKT * kSyrkRow;
    double* const Ag = smem + g * 4 * kGStage;
    double4_t acc[T::NT];
#pragma unroll
    for (int n = 0; n < T::NT; ++n)
        acc[n] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
    const int n0 = (kt1 - kt0 + 1) / 2;            // stages of half 0 (the longer one)
    const int my_n = (kt1 - kt0 + 1 - g) / 2;      // stages of this half: kt0 + g, kt0 + g + 2, ...
    double4_t va[2], vb[2];
    auto gload = [&](const int i) {
        const int kt = kt0 + 2 * i + g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double* zr = Z + (int64_t)(kt * kSyrkKT + lr + 8 * h) * ldz;
            va[h] = *reinterpret_cast<const double4_t*>(zr + I0 + lc);
            if (!DIAG)   // a diagonal tile's two operands are the same columns of Z: one copy in the LDS
                vb[h] = *reinterpret_cast<const double4_t*>(zr + J0 + lc);
        }
    };
    auto lstore = [&](const int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<double4_t*>(&Ag[buf * 2 * kGStage + (lr + 8 * h) * kSyrkRow + lc]) = va[h];
            if (!DIAG)
                *reinterpret_cast<double4_t*>(&Ag[buf * 2 * kGStage + kGStage + (lr + 8 * h) * kSyrkRow + lc]) = vb[h];
        }
    };
    struct Frag {
        double a[T::NA], b[T::NB];
    };
    auto read_ops = [&](const int buf, const int ks, Frag& f) {
        const double* Ap = Ag + buf * 2 * kGStage;
        const double* Bp = DIAG ? Ap : Ap + kGStage;
        const int row = (ks * 4 + fk) * kSyrkRow;
#pragma unroll
        for (int i = 0; i < T::NA; ++i)
            f.a[i] = Ap[row + 16 * T::arow(i) + fi];
#pragma unroll
        for (int j = 0; j < T::NB; ++j)
            f.b[j] = Bp[row + 16 * T::bcol(j) + fi];
    };
    auto mfma_step = [&](const Frag& f) {
#pragma unroll
        for (int n = 0; n < T::NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[T::ai(n)], f.b[T::bi(n)], acc[n], 0, 0, 0);
    };
    // The two halves run HALF A STAGE APART.  Left in step, all eight waves reach the end of a stage together: 64 KB of
    // ds_write, the barrier, 32 KB of ds_read before anybody's first MFMA -- ~1200 idle cycles of the matrix pipe per
    // 8192 (measured as 80 us in the loop for 70 us of MFMAs).  Here every barrier is a stage boundary for one half and
    // the middle of a stage for the other, which has the operands of its next K step in registers and keeps the pipe of the
    // SIMD they share busy while the first half writes, waits and reads.  Half 1 starts one barrier late and half 0 ends
    // one barrier early; both execute 2 n0 + 2 barriers.
    if (my_n > 0) {
        gload(0);
        lstore(0);
    }
    if (g == 1)
        __syncthreads();
    for (int i = 0; i < n0; ++i) {
        __syncthreads();   // this half's stage i is in its buffer; the other half is in the middle of a stage
        const bool have = i < my_n, more = i + 1 < my_n;
        const int buf = i & 1;
        Frag f0, f1, f2;
        if (more)
            gload(i + 1);   // the next stage's loads fly while this stage's MFMAs issue
        if (have) {
            read_ops(buf, 0, f0);
            read_ops(buf, 1, f1);
            mfma_step(f0);
            read_ops(buf, 2, f2);   // in registers before the barrier: the first MFMAs behind it do not wait for the LDS
            mfma_step(f1);
        }
        __syncthreads();   // the middle of this half's stage; a stage boundary of the other half
        if (have) {
            mfma_step(f2);
            read_ops(buf, 3, f0);
            mfma_step(f0);
        }
        if (more)
            lstore(buf ^ 1);
    }
    if (g == 0)
        __syncthreads();
    __syncthreads();   // every MFMA operand is read: the stage buffers are dead
    // The two halves exchange half of their accumulators through the LDS (dead now: the loop ends with a barrier); each
    // then holds the sums of half of the wave's MFMA tiles (own + other: the same bits either way) and stores those.
    constexpr int H = T::NT / 2;
    double* const X = smem + (size_t)W * 16 * 256;   // [wave of the half][tile 0..15][4][64]
#pragma unroll
    for (int n = 0; n < T::NT; ++n)
        if ((n < H) == (g == 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                X[(n * 4 + r) * 64 + lane] = acc[n][r];
        }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < T::NT; ++n)
        if ((n < H) == (g == 0)) {
            const int tr = T::arow(T::ai(n)), tc = T::bcol(T::bi(n));
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[n][r] + X[(n * 4 + r) * 64 + lane];
                Cb[(16 * tr + fk + 4 * r) * kSyrkT + 16 * tc + fi] = v[r];
            }
            if (DIAG && tr > tc && (tr >> 2) == (tc >> 2)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cb[(16 * tc + fi) * kSyrkT + 16 * tr + fk + 4 * r] = v[r];
            }
        }
}

__global__ __launch_bounds__(512, 1) void k_syrk_wide(const LmCtl* ctl, const double* __restrict__ Z, int ldz, SyrkPlanDev pl)
{
    if (ctl && ctl->done)
        return;
    __shared__ __attribute__((aligned(16))) double smem[4 * kWideStage];   // 144 KB
    static_assert(4 * kWideStage >= 4 * 16 * 256, "the accumulator exchange must fit into the stage buffers");
    const int gidx = blockIdx.x;
    const int64_t u = pl.wg_u0[gidx], u_end = pl.wg_u1[gidx];
    if (u >= u_end)
        return;
    const int t = (int)(u / pl.n_kt);
    const int kt0 = (int)(u % pl.n_kt), kt1 = kt0 + (int)(u_end - u);   // one tile per workgroup (host plan)
    const int I0 = pl.tile_bi[t] * kSyrkT, J0 = pl.tile_bj[t] * kSyrkT;
    double* Cb = pl.partials + (size_t)pl.wg_seg0[gidx] * kSyrkTile;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 3);
    if (I0 == J0) {
        switch (w) {
        case 0: syrk_wide_item<true, 0>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        case 1: syrk_wide_item<true, 1>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        case 2: syrk_wide_item<true, 2>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        default: syrk_wide_item<true, 3>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        }
    } else {
        switch (w) {
        case 0: syrk_wide_item<false, 0>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        case 1: syrk_wide_item<false, 1>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        case 2: syrk_wide_item<false, 2>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        default: syrk_wide_item<false, 3>(Z, ldz, I0, J0, kt0, kt1, smem, Cb); break;
        }
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
                                                         double* __restrict__ S, DiagArgs da,
                                                         double* __restrict__ packed = nullptr)
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
        if (!ADD_DIAG && packed) {
            // world > 1: straight into the packed lower triangle that travels through the all-reduce (k_pack_lower's
            // layout: row r holds its r + 1 leading entries)
            const int64_t base = (int64_t)row * (row + 1) / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (J0 + tc + h <= row)
                    packed[base + J0 + tc + h] = out[h];
        } else {
            *reinterpret_cast<double2*>(S + (int64_t)row * ld + J0 + tc) = make_double2(out[0], out[1]);
        }
    }
}

// ---- block-sparse reduced system: S(f, f') -= sum over the eliminated poses e that see both f and f' --------------
//
// The dense path stores Z with its zero blocks and multiplies them (at 25 % visibility 15/16 of the rank-k flops);
// here Z holds only the blocks of co-observed pairs (Engine::Zc, one column-major 6x6 block per observation, E order)
// and the product runs over pairs of observations that share an eliminated pose -- what the sparse normal-Cholesky
// behind ceres::Solve (src/TagReconstructor.cpp:725-738) exploits on projects where an image sees a handful of tags
// (README.md:155-216).  The structure is symbolic work done once at vmm_ba_create: for every pair of kept poses
// (f, f' <= f) the list of its terms (position of observation (e, f) in f's row, E-order index of (e, f')), in e order.
//   * a lane owns one output column (f, f', c): six accumulators in registers, summed over the pair's terms in list
//     order -- no atomics, no shared accumulator, one writer per element: bit-repeatable;
//   * a workgroup takes up to 42 consecutive pairs of ONE row f, so the left operands Z_ef (the row's blocks, 288 B
//     each) are staged once in LDS (128 at a time) and read from there; the right operand is six doubles per term;
//   * every pair of the lower triangle is written, the empty ones as zeros, together with the kept family's damped
//     diagonal block; each row's last pair is its right-hand side entry b_f = s_f g_f - sum_e Z_ef^T z_e.
struct PairArgs {
    const LmCtl* ctl;
    const int32_t* item_row;    // [n_items] kept pose of the item
    const int32_t* item_p0;     // [n_items] first pair (global pair id); the item ends at min(p0 + kPairsPerItem, end of the row)
    const int32_t* pair_start;  // [n_f + 1] first pair id of every row; row f has the pairs f' = 0..f and the rhs
    const int32_t* tstart;      // [n_pairs + 1] term range of every pair
    const int2* terms;          // [n_terms] .x: position of the left block in the row's observation list, .y: E-order
                                // observation index of the right block (rhs pair: the eliminated pose)
    const int32_t* f_start;     // [n_f + 1] F-order observation range of every kept pose
    const int32_t* f2e;         // [n_obs] F order -> E-order index
    const double* Zc;           // [n_obs][36]
    const double* ze;           // [n_e][6]
    double* S;
    int ld, n_items, n_f;
    DiagArgs da;
    int add_diag;
    const int32_t* pair_col;    // explicit form: [n_pairs] first column of the pair's block (-1: right-hand side); else null
    const int32_t* row_of;      // explicit form: [n_f] first row of kept pose f in S
};

#ifndef VMM_PAIR_NS
#define VMM_PAIR_NS 2
#endif
constexpr int kPairChunk = 128;   // left blocks staged per pass: 38 KB of LDS, four workgroups per CU
constexpr int kPairSplit = VMM_PAIR_NS;                  // lanes that share one output column: each takes every
                                                         // kPairSplit-th term, the partial sums meet in a fixed shuffle tree
#ifndef VMM_PAIR_THREADS
#define VMM_PAIR_THREADS 256
#endif
constexpr int kPairThreads = VMM_PAIR_THREADS;           // workgroup size of k_schur_pairs
constexpr int kPairsPerItem = kPairThreads / (6 * kPairSplit);    // 6 x kPairSplit lanes per pair
constexpr int kPairBlk = 38;      // doubles per staged block: 36 + 2, so that consecutive blocks start 12 banks apart

__global__ __launch_bounds__(kPairThreads) void k_schur_pairs(PairArgs a)
{
    if (a.ctl && a.ctl->done)
        return;
    const int tid = threadIdx.x;
    const int n_red = a.da.n_red, n_pad = a.da.n_pad, ld = a.ld;
    if ((int)blockIdx.x >= a.n_items) {
        // what no pair writes: the rest of every row's diagonal 64-block (the Cholesky kernels load whole 16x16 tiles
        // of it) and the padding of the reduced system (unit rows on one GPU; world > 1: zero, k_unpack_diag sets the ones
        // behind the all-reduce)
        const int nb = gridDim.x - a.n_items, b = (int)blockIdx.x - a.n_items;
        for (int row = b; row <= n_pad; row += nb) {
            if (row == n_pad) {
                for (int j = n_red + tid; j < n_pad; j += kPairThreads)
                    a.S[(int64_t)n_pad * ld + j] = 0.0;
            } else if (row >= n_red) {
                for (int j = tid; j < n_pad; j += kPairThreads)
                    a.S[(int64_t)row * ld + j] = (a.add_diag && j == row) ? 1.0 : 0.0;
            } else {
                const int c0 = 6 * (row / 6) + 6, c1 = min((row / 64) * 64 + 64, n_pad);
                for (int j = c0 + tid; j < c1; j += kPairThreads)
                    a.S[(int64_t)row * ld + j] = 0.0;
            }
        }
        return;
    }
    __shared__ __attribute__((aligned(16))) double As[kPairChunk * kPairBlk];
    const int f = a.item_row[blockIdx.x];
    const int row_p0 = a.pair_start[f];
    const int sl = tid % kPairSplit;               // which of the pair's term subsequences
    const int p = a.item_p0[blockIdx.x] + tid / (6 * kPairSplit), c = (tid / kPairSplit) % 6;
    const int j = p - row_p0;                      // implicit form: f' (0..f) or f + 1: the right-hand side
    const bool expl = a.pair_col != nullptr;
    const bool valid = tid < 6 * kPairSplit * kPairsPerItem && (expl ? p < a.pair_start[f + 1] : j <= f + 1);
    const int rbase = expl ? a.row_of[f] : 6 * f;                    // first row of f's blocks in S
    const int cbase = expl ? (valid ? a.pair_col[p] : 0) : 6 * j;    // first column of this pair's block
    const bool rhs = expl ? cbase < 0 : j == f + 1;
    int t = valid ? a.tstart[p] + sl : 0;
    const int t1 = valid ? a.tstart[p + 1] : 0;
    const int fs = a.f_start[f], len = a.f_start[f + 1] - fs;
    const double* __restrict__ Zc = a.Zc;
    double acc[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    const bool active = valid && (!rhs || c == 0);
    const int2* __restrict__ terms = a.terms;
    for (int lo = 0; lo < len; lo += kPairChunk) {
        const int hi = min(lo + kPairChunk, len);
        if (lo > 0)
            __syncthreads();   // the previous pass's blocks are no longer read
        // stage the row's blocks lo .. hi-1: thread -> (block, pair of doubles).  All indices first, then all blocks:
        // two memory latencies per pass, not two per block.
        {
            constexpr int NI = (kPairChunk * 18 + kPairThreads - 1) / kPairThreads;
            int ie[NI];
            double2 v[NI];
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = tid + kPairThreads * u;
                const int k = min(i / 18, hi - lo - 1);
                ie[u] = a.f2e[fs + lo + k];
            }
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = tid + kPairThreads * u;
                v[u] = *reinterpret_cast<const double2*>(Zc + 36 * (int64_t)ie[u] + 2 * (i % 18));
            }
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = tid + kPairThreads * u;
                if (i < (hi - lo) * 18)
                    *reinterpret_cast<double2*>(&As[kPairBlk * (i / 18) + 2 * (i % 18)]) = v[u];
            }
        }
        __syncthreads();
        // The pair's terms of this pass, eight at a time: the eight (left, right) index pairs are loaded together, then
        // the 48 right-operand values, so a batch pays two memory latencies instead of three per term (a lane walking
        // its list term by term spends its time waiting: measured 115 us per launch at 500 x 200, 25 % visibility).
#ifndef VMM_PAIR_TB
#define VMM_PAIR_TB 4
#endif
        constexpr int TB = VMM_PAIR_TB;
        // (the index pairs of the batch after this one are requested behind this batch's right operands, before its
        // multiply-adds: one memory round trip per batch instead of two -- the waves of this kernel spend most of their
        // life waiting, PMC SQ_WAIT_ANY 63 %)
        int2 tt[TB];
        if (active && t < t1) {
#pragma unroll
            for (int u = 0; u < TB; ++u)
                tt[u] = terms[min(t + u * kPairSplit, t1 - 1)];
        }
        while (active && t < t1) {
            double b[TB][6];
            bool ok[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                ok[u] = t + u * kPairSplit < t1 && tt[u].x < hi;   // a prefix: the terms are ordered by left position
                if (rhs) {
#pragma unroll
                    for (int r = 0; r < 6; ++r)
                        b[u][r] = a.ze[6 * (int64_t)tt[u].y + r];
                } else {
                    // column c of the right block: 48 contiguous, 16-byte aligned bytes
                    const double2* __restrict__ bp = reinterpret_cast<const double2*>(Zc + 36 * (int64_t)tt[u].y + 6 * c);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double2 v2 = bp[r];
                        b[u][2 * r] = v2.x;
                        b[u][2 * r + 1] = v2.y;
                    }
                }
            }
            int2 tn[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u)
                tn[u] = terms[min(t + (TB + u) * kPairSplit, t1 - 1)];
            int n_ok = 0;
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                if (ok[u]) {
                    const double* __restrict__ Ap = As + kPairBlk * (tt[u].x - lo);
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            acc[q] = fma(Ap[6 * q + r], b[u][r], acc[q]);   // A(r, q), column-major
                    ++n_ok;
                }
            }
            t += n_ok * kPairSplit;
            if (n_ok < TB)
                break;   // the rest of the list belongs to a later pass (or the list is finished)
#pragma unroll
            for (int u = 0; u < TB; ++u)
                tt[u] = tn[u];
        }
    }
    // the kPairSplit partial sums of a column sit in neighbouring lanes: fixed tree
#pragma unroll
    for (int m = 1; m < kPairSplit; m <<= 1)
#pragma unroll
        for (int q = 0; q < 6; ++q)
            acc[q] += __shfl_xor(acc[q], m, 64);
    if (!active || sl != 0)
        return;
    if (a.add_diag && a.ctl) {
        a.da.H_F += small_sel(a.ctl, a.da.alt_off);
        a.da.g_F += small_sel(a.ctl, a.da.alt_off);
    }
    // (scale_F, g_F, D2_F, H_F are indexed by the kept pose's parameter number 6 f + q; S by its row rbase + q)
    if (rhs) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int par = 6 * f + q;
            a.S[(int64_t)n_pad * ld + rbase + q] = (a.add_diag ? a.da.scale_F[par] * a.da.g_F[par] : 0.0) - acc[q];
        }
        return;
    }
    const bool diag_pair = cbase == rbase;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int par = 6 * f + q;
        double v = -acc[q];
        if (a.add_diag && diag_pair) {
            v += a.da.scale_F[par] * a.da.H_F[36 * (int64_t)f + 6 * q + c] * a.da.scale_F[6 * f + c];
            if (q == c)
                v += a.da.D2_F[par];
        }
        a.S[(int64_t)(rbase + q) * ld + cbase + c] = v;
    }
}

// Explicit pair lists write only the blocks that exist: everything else the factorisation reads -- the lower block
// triangle up to the end of each row's diagonal 64-block, the right-hand side row -- is zeroed first, with a unit diagonal
// (one GPU; the pairs overwrite it on the rows of kept poses, it stays on the padding rows).
__global__ __launch_bounds__(256) void k_fill_lower(const LmCtl* ctl, double* __restrict__ S, int ld, int n_pad, int unit_diag)
{
    if (ctl && ctl->done)
        return;
    const int row = blockIdx.x;   // 0 .. n_pad
    const int c1 = row == n_pad ? n_pad : (row / 64) * 64 + 64;
    double2* __restrict__ dst = reinterpret_cast<double2*>(S + (int64_t)row * ld);
    for (int j = threadIdx.x; 2 * j < c1; j += 256) {
        double2 v = make_double2(0.0, 0.0);
        if (unit_diag && row < n_pad) {
            if (2 * j == row)
                v.x = 1.0;
            if (2 * j + 1 == row)
                v.y = 1.0;
        }
        dst[j] = v;
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
    if (p.n_wg > 0 && p.wide)
        hipLaunchKernelGGL(k_syrk_wide, dim3(p.n_wg), dim3(512), 0, st, ctl, Z, ldz, plan_dev(p));
    else if (p.n_wg > 0)
        hipLaunchKernelGGL(k_syrk_streamk, dim3(p.n_wg), dim3(256), 0, st, ctl, Z, ldz, plan_dev(p));
}

void launch_reduce_plan(hipStream_t st, const LmCtl* ctl, const SyrkPlan& p, int ld, int n_rows, double* S)
{
    DiagArgs da = {};
    if (p.n_tiles > 0)
        hipLaunchKernelGGL((k_reduce_partials<false>), dim3(16 * p.n_tiles), dim3(256), 0, st, ctl, plan_dev(p), ld, n_rows,
                           S, da);
}

int schur_pairs_per_item() { return kPairsPerItem; }

void launch_schur_rows(Engine& e, bool add_diag)
{
    if (e.n_row_items <= 0)
        return;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    PairArgs a;
    a.ctl = e.ctl;
    a.item_row = e.row_items;
    a.item_p0 = e.row_items + e.n_row_items;
    a.pair_start = e.pair_start;
    a.tstart = e.pair_tstart;
    a.terms = reinterpret_cast<const int2*>(e.pair_terms);
    a.f_start = e.ordF.start;
    a.f2e = e.f2e;
    a.Zc = e.Zc;
    a.ze = e.ze;
    a.S = e.S;
    a.ld = e.ldz;
    a.n_items = e.n_row_items;
    a.n_f = e.n_f;
    a.da.H_F = e.elim_cams ? e.H_tag : e.H_cam;
    a.da.g_F = e.elim_cams ? e.g_tag : e.g_cam;
    a.da.scale_F = e.scale + 6 * (size_t)f_off;
    a.da.D2_F = e.D2 + 6 * (size_t)f_off;
    a.da.n_red = e.n_red;
    a.da.n_pad = e.n_pad;
    a.da.alt_off = e.small_alt_off;
    a.add_diag = add_diag ? 1 : 0;
    a.pair_col = e.explicit_pairs ? e.pair_col : nullptr;   // (launch_schur_rows only runs on the block-sparse path)
    a.row_of = e.explicit_pairs ? e.row_of : nullptr;
    if (e.explicit_pairs)
        hipLaunchKernelGGL(k_fill_lower, dim3(e.n_pad + 1), dim3(256), 0, e.stream, (const LmCtl*)e.ctl, e.S, e.ldz, e.n_pad,
                           add_diag ? 1 : 0);
    // implicit form: extra workgroups for the rows' zero fill and the padding
    const int n_fill = e.explicit_pairs ? 0 : std::min(64, e.n_pad + 1);
    hipLaunchKernelGGL(k_schur_pairs, dim3(e.n_row_items + n_fill), dim3(kPairThreads), 0, e.stream, a);
}

void launch_syrk_only(Engine& e)
{
    if (e.sparse_schur)
        launch_schur_rows(e, !e.multi);
    else
        launch_syrk_plan(e.stream, e.ctl, e.Z, e.ldz, e.syrk);
}

// S = -(sum of partials) [+ damped diagonal blocks and rhs on one GPU; with world > 1 they are added by
// k_unpack_diag after the all-reduce, being identical on every rank]
void launch_syrk_reduced(Engine& e)
{
    launch_syrk_only(e);
    if (e.sparse_schur)
        return;   // k_schur_rows writes S itself (with the diagonal blocks on one GPU)
    if (e.multi) {
        DiagArgs none = {};
        if (e.syrk.n_tiles > 0)
            hipLaunchKernelGGL((k_reduce_partials<false>), dim3(16 * e.syrk.n_tiles), dim3(256), 0, e.stream, e.ctl,
                               plan_dev(e.syrk), e.ldz, e.n_pad + 1, e.S, none, e.S_packed);
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

// Behind the all-reduce: row r of the packed triangle back into S, plus what is identical on every rank and therefore
// not part of the sum: the kept family's damped diagonal blocks S_ff += s H_f s + D_f^2, the right-hand side row
// += s_f g_f, and the unit diagonal of the padding.
__global__ __launch_bounds__(256) void k_unpack_diag(const LmCtl* ctl, const double* __restrict__ packed, int ld,
                                                     int f_off_pose, const double* __restrict__ H_F,
                                                     const double* __restrict__ g_F, const double* __restrict__ scale,
                                                     const double* __restrict__ D2, double* __restrict__ S, int n_red,
                                                     int n_pad, const int32_t* __restrict__ pose_of_row,
                                                     const int32_t* __restrict__ row_of)
{
    if (ctl && ctl->done)
        return;
    const int r = blockIdx.x;
    const int64_t base = (int64_t)r * (r + 1) / 2;
    const double* sF = scale + 6 * (int64_t)f_off_pose;
    const double* dF = D2 + 6 * (int64_t)f_off_pose;
    if (pose_of_row) {
        // tree ordering of the kept family: pose f's rows are row_of[f] .. + 5, every other row is padding
        const int fr = r < n_pad ? pose_of_row[r] : -1;
        for (int c = threadIdx.x; c <= r; c += 256) {
            double v = packed[base + c];
            const int fc = c < n_pad ? pose_of_row[c] : -1;
            if (r == n_pad) {
                if (fc >= 0) {
                    const int k = 6 * fc + (c - row_of[fc]);
                    v += sF[k] * g_F[k];
                }
            } else if (fr >= 0) {
                if (fc == fr) {
                    const int a = r - row_of[fr], b = c - row_of[fr];
                    double add = sF[6 * fr + a] * H_F[36 * (int64_t)fr + 6 * a + b] * sF[6 * fr + b];
                    if (c == r)
                        add += dF[6 * fr + a];
                    v += add;
                }
            } else if (c == r) {
                v = 1.0;
            }
            S[(int64_t)r * ld + c] = v;
        }
        return;
    }
    for (int c = threadIdx.x; c <= r; c += 256) {
        double v = packed[base + c];
        if (r == n_pad) {
            if (c < n_red)
                v += sF[c] * g_F[c];
        } else if (r < n_red) {
            if (c / 6 == r / 6) {
                double add = sF[r] * H_F[36 * (int64_t)(r / 6) + 6 * (r % 6) + c % 6] * sF[c];
                if (c == r)
                    add += dF[r];
                v += add;
            }
        } else if (c == r) {
            v = 1.0;
        }
        S[(int64_t)r * ld + c] = v;
    }
}

// World > 1, in front of the all-reduce.  The dense rank-k path packs in its reduction kernel (launch_syrk_reduced);
// only the block-sparse path, whose kernel writes S itself, needs the copy.
void launch_pack_lower(Engine& e, bool unpack)
{
    if (!e.multi)
        return;
    if (!unpack) {
        if (e.sparse_schur)
            hipLaunchKernelGGL(k_pack_lower, dim3(e.n_pad + 1), dim3(256), 0, e.stream, (const LmCtl*)e.ctl,
                               (const double*)e.S, e.ldz, e.n_pad + 1, e.S_packed, false, e.S);
        return;
    }
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const double* H_F = e.elim_cams ? e.H_tag : e.H_cam;
    const double* g_F = e.elim_cams ? e.g_tag : e.g_cam;
    hipLaunchKernelGGL(k_unpack_diag, dim3(e.n_pad + 1), dim3(256), 0, e.stream, (const LmCtl*)e.ctl,
                       (const double*)e.S_packed, e.ldz, f_off, H_F, g_F, (const double*)e.scale, (const double*)e.D2,
                       e.S, e.n_red, e.n_pad, (const int32_t*)e.pose_of_row, (const int32_t*)e.row_of);
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
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_schur_pairs)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_fill_lower)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_syrk_streamk)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_syrk_wide)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_reduce_partials<true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_reduce_partials<false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_unpack_diag)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_pack_lower)) != hipSuccess;
    return bad;
}

} // namespace vmm
