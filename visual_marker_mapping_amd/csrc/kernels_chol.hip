// Dense Cholesky factorisation + triangular solves of the reduced system (gfx950).
//
// Replaces the numeric Cholesky of the linear solver Ceres runs inside ceres::Solve for the reference
// (src/TagReconstructor.cpp:737-738; SPARSE_NORMAL_CHOLESKY / DENSE_QR by default, both exact).
//
// Matrix layout: row-major, leading dimension ld, only the lower block triangle is used.  Order
// n_pad = 64 * n_blk; the right-hand side is stored as ROW n_pad of the same array, so the blocked
// right-looking factorisation also performs the forward substitution (row n_pad ends as (L^-1 b)^T).
//
// One launch per block column k (k_chol_step) holds three kinds of workgroups:
//   * panel: the tall block [A_kk; A_ik] is factored in MFMA accumulators, eight columns per round (the
//     8x8 pivot block in registers, rows scaled, rank-8 MFMA update); every panel workgroup re-factors the
//     64x64 diagonal block (cheaper than a dependent launch) and owns 64 rows below it.  It first applies
//     the rank-64 update of its own block column from panel k-1 (look-ahead), and stores its rows of L to S
//     and, transposed, to P (64 x ld);
//   * update: the trailing update A_ij -= L_i,k-1 L_j,k-1^T of panel k-1 for the block columns >= k+1, from
//     the transposed panel of launch k-1 (k-major operands for v_mfma_f64_16x16x4_f64); the workgroups loop
//     over the 64x64 tiles with the next tile's operands requested ahead;
//   * one workgroup inverts the 64x64 diagonal factor of block k-1 for the back-substitution.
// The back-substitution L^T y = w is ONE launch of n_blk workgroups handing their 64 unknowns on through
// self-validating granules (k_backsolve_chain); k_backsolve_step is the per-block fallback.
#include <type_traits>
#include <utility>

#include "engine.hpp"

namespace vmm {

#ifdef VMM_STAMPS
__device__ unsigned long long g_stamps[64];
#define STAMP(slot)                                                                  \
    do {                                                                             \
        if (blockIdx.x == 1 && threadIdx.x == 0 && k == 1) {                         \
            g_stamps[slot] = __builtin_amdgcn_s_memtime();                           \
            g_stamps[16 + slot] = __builtin_amdgcn_s_memrealtime();                  \
        }                                                                            \
    } while (0)
#define USTAMP(slot)                                                                 \
    do {                                                                             \
        if (u == 0 && threadIdx.x == 0 && k == 1 && t == u + n_wg)                   \
            g_stamps[(slot)] = __builtin_amdgcn_s_memtime();                         \
    } while (0)
#else
#define USTAMP(slot)
#define STAMP(slot)
#endif
#ifdef VMM_STAMPS
#define RSTAMP(slot)                                                                 \
    do {                                                                             \
        if (blockIdx.x == 1 && threadIdx.x == 0 && J0 == 32 && g_stamps[slot] == 0)  \
            g_stamps[slot] = __builtin_amdgcn_s_memtime();                           \
    } while (0)
#else
#define RSTAMP(slot)
#endif

} // namespace vmm
#include "potrf64.inc"
namespace vmm {

// NOTE on the diagonal factor: L_kk goes to its own buffer Ld[k][64][64], never back into S(k,k): every
// workgroup of the launch reads S(k,k) when it starts, and a workgroup that starts late (busy GPU, more
// workgroups than CUs) must still find the unfactored block there.
//
// Panel of block column k as ONE right-looking factorisation of the tall matrix [A_kk; A_ik]:
// workgroup 0 owns only the diagonal block, workgroup b >= 1 the diagonal block (re-factored
// redundantly, cheaper than a dependent launch) plus 64 rows below it (the rhs row n_pad is just one
// more row).  Both 64x64 blocks live in v_mfma_f64_16x16x4_f64 accumulators for the whole kernel:
// wave w holds the 16-row tile row w (tiles (w,0..3); for the diagonal block only tj <= w).
// Eight rounds of eight columns:
//   1. the lanes that hold columns J0..J0+7 publish them to a small LDS panel buffer
//   2. wave 0 (diagonal rows) and wave 1 (rows below) each factor the 8x8 pivot block in registers
//      (eight dependent rsqrt chains, no barrier in between) and scale "their" row: x = a L8^{-T},
//      written back in place
//   3. every wave applies the rank-8 update C -= X X_d^T to its tiles with two MFMAs per tile
// The panel buffers ping-pong between rounds, so two barriers per round suffice and there is no
// separate triangular-solve phase: after the last round the scaled columns ARE L_ik.
// (Four columns per round cost 16 x (2 barriers + 2 LDS round trips); eight halve that overhead for
// the same pivot chain.)
// block structure of a tree-ordered factor (DfArgs::nz): bit k of block row i
__device__ __forceinline__ bool nz_bit(const unsigned long long* nz, const int i, const int k)
{
    return (nz[kDfMaskWords * i + (k >> 6)] >> (k & 63)) & 1ull;
}

constexpr int kPs = 9;   // LDS row stride (doubles) of the 64x8 panel buffers: conflict-free rows
constexpr int kPw = 8;   // columns per round

__device__ __forceinline__ constexpr int tri8(int r, int c) { return r * (r + 1) / 2 + c; }

struct Piv8 {
    double l[36];     // lower triangle of the 8x8 factor, packed row-major (diagonal included)
    double inv[8];    // reciprocals of its diagonal
    bool ok;
};

// Cholesky of the symmetric 8x8 block at D (LDS, row stride kPs, lower triangle), in registers.
// Right-looking: as soon as column j is scaled, its outer product is subtracted from the columns to its right, so
// the NEXT pivot depends on one multiply and one fused multiply-add behind the reciprocal square root instead of
// on a j-deep chain of dependent FMAs (the left-looking form cost ~143 cycles per pivot, this one ~100); the other
// updates are independent and fill the issue slots the chain leaves free.
__device__ __forceinline__ void chol8(const double* __restrict__ D, Piv8& p)
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c)
            p.l[tri8(r, c)] = D[r * kPs + c];
    p.ok = true;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double t = p.l[tri8(j, j)];
        bool okj = true;
        const double inv = safe_rsqrt(t, okj);
        p.ok = p.ok && okj;
        p.inv[j] = inv;
        p.l[tri8(j, j)] = t * inv;
#pragma unroll
        for (int i = j + 1; i < 8; ++i)
            p.l[tri8(i, j)] *= inv;
        // the next pivot's diagonal first
#pragma unroll
        for (int c = j + 1; c < 8; ++c)
#pragma unroll
            for (int i = c; i < 8; ++i)
                p.l[tri8(i, c)] = fma(-p.l[tri8(i, j)], p.l[tri8(c, j)], p.l[tri8(i, c)]);
    }
}

// x <- x L8^{-T} (a row of eight columns scaled by the pivot block's factor), right-looking for the same reason:
// every step is one multiply behind the previous step's update instead of a q-deep chain.
__device__ __forceinline__ void scale8(double (&x)[8], const Piv8& p)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        x[q] *= p.inv[q];
#pragma unroll
        for (int c = q + 1; c < 8; ++c)
            x[c] = fma(-x[q], p.l[tri8(c, q)], x[c]);
    }
}

template <int J0, bool HAS_T>
__device__ __forceinline__ void panel_round(const int w, const int lane,
                                            double4_t (&Dacc)[4], double4_t (&Tacc)[4], double* __restrict__ Pd,
                                            double* __restrict__ Pt, double* __restrict__ At,
                                            double* __restrict__ R, double* __restrict__ invd, bool& ok)
{
    constexpr int tc = J0 >> 4, cj = J0 & 15;
    const int fr = lane & 15, fk = lane >> 4;
    double* pd = Pd + ((J0 >> 3) & 1) * 64 * kPs;
    double* pt = Pt + ((J0 >> 3) & 1) * 64 * kPs;
    // 1. publish columns J0..J0+7 (rows of my tile row) from the accumulators
    if (fr >= cj && fr < cj + kPw) {
        const int q = fr - cj;
        const int row = 16 * w + fk;
        if (w >= tc) {
            pd[(row + 0) * kPs + q] = Dacc[tc][0];
            pd[(row + 4) * kPs + q] = Dacc[tc][1];
            pd[(row + 8) * kPs + q] = Dacc[tc][2];
            pd[(row + 12) * kPs + q] = Dacc[tc][3];
        }
        if (HAS_T) {
            pt[(row + 0) * kPs + q] = Tacc[tc][0];
            pt[(row + 4) * kPs + q] = Tacc[tc][1];
            pt[(row + 8) * kPs + q] = Tacc[tc][2];
            pt[(row + 12) * kPs + q] = Tacc[tc][3];
        }
    }
    __syncthreads();
    // 2. pivot block + row scaling (wave 0: diagonal rows, wave 1: rows below)
    if (w == 0 || (w == 1 && HAS_T)) {
        double* row = (w == 0 ? pd : pt) + lane * kPs;
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            x[q] = row[q];
        Piv8 p;
        chol8(pd + J0 * kPs, p);
        scale8(x, p);   // x = a L8^{-T}
        if (w == 0) {
            ok = ok && p.ok;
            const int r = lane - J0;
            const bool below = r >= kPw, above = r < 0;
            if (below) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    row[q] = x[q];
            }
            if (!HAS_T) {
                // keep L^T for the write-back: x below the pivot block, the factor inside, zero above
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    double inside = 0.0;
#pragma unroll
                    for (int rr = q; rr < 8; ++rr)
                        inside = (r == rr) ? p.l[tri8(rr, q)] : inside;
                    At[(J0 + q) * kLdT + lane] = below ? x[q] : (above ? 0.0 : inside);
                }
                if (r >= 0 && r < kPw) {
                    double iv = 0.0;
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr)
                        iv = (r == rr) ? p.inv[rr] : iv;
                    invd[lane] = iv;
                }
            }
        } else {
            double* rr = R + lane * kLd + J0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                row[q] = x[q];
                rr[q] = x[q];
            }
        }
    }
    __syncthreads();
    // 3. rank-8 update of the tiles right of the pivot columns.  MFMA f64 maps: A[i = lane&15][k = lane>>4],
    //    B[k = lane>>4][j = lane&15], C row = (lane>>4) + 4*reg, col = lane&15.
    if (J0 + kPw < 64) {
        constexpr int t0 = (J0 + kPw) >> 4;
        const int ra = 16 * w + fr;
        const bool ma = ra >= J0 + kPw;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const double adv = pd[ra * kPs + 4 * ks + fk];
            const double ad = ma ? -adv : 0.0;
            const double at = HAS_T ? -pt[ra * kPs + 4 * ks + fk] : 0.0;
            // the tile that holds the next pivot columns goes first: the next round's publish waits on it
#pragma unroll
            for (int tj = t0; tj < 4; ++tj) {
                const int rb = 16 * tj + fr;
                const double bv = pd[rb * kPs + 4 * ks + fk];
                const double b = (rb >= J0 + kPw) ? bv : 0.0;
                // tiles above the diagonal (tj > w) get a zero operand instead of a branch
                const double adm = (tj <= w) ? ad : 0.0;
                Dacc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(adm, b, Dacc[tj], 0, 0, 0);
                if (HAS_T)
                    Tacc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, b, Tacc[tj], 0, 0, 0);
            }
        }
    }
}

template <bool HAS_T>
__device__ __forceinline__ void panel_body(LmCtl* ctl, double* __restrict__ S, int ld, int n_pad, int k,
                                           double* __restrict__ P, const double* __restrict__ Pprev,
                                           const double* __restrict__ Pprev2, double* __restrict__ dinv,
                                           double* __restrict__ Ld, double* RA, double* Pd, double* Pt, double* invd,
                                           double* Ads, double* Ats)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int K0 = k * kNB;
    const int R0 = K0 + kNB + ((int)blockIdx.x - 1) * 64;
    STAMP(0);
    // accumulator-layout loads straight from global memory: for fixed (tile, reg) 16 lanes read 128
    // contiguous bytes of one row
    double4_t Dacc[4], Tacc[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        Dacc[tj] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
        Tacc[tj] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + fk + 4 * r;
            const double dv = S[(int64_t)(K0 + row) * ld + K0 + 16 * tj + fr];
            Dacc[tj][r] = (tj <= w) ? dv : 0.0;
            if (HAS_T) {
                const int grow = (R0 + row <= n_pad) ? R0 + row : n_pad;   // clamp: always in bounds
                const double tv = S[(int64_t)grow * ld + K0 + 16 * tj + fr];
                Tacc[tj][r] = (R0 + row <= n_pad) ? tv : 0.0;
            }
        }
    }
    // Look-ahead: the trailing updates skip this block column (chol_update2_wg starts one or two columns further), so
    // this kernel does not have to wait for them; the missing rank-64 updates of the tiles (k,k) and (i,k) are applied
    // here from the transposed panels that are still pending: Pprev2 (block column k-2; even k only, the pair of
    // panels k-2, k-1 is applied to the rest of the matrix by this launch and the next) and Pprev (k-1).
    STAMP(6);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const double* __restrict__ Pq = pp == 0 ? Pprev2 : Pprev;
        if (!Pq)
            continue;
        if (pp == 1 && Pprev2)
            __syncthreads();   // the first panel's operands are consumed
        // stage Pq[:, K0..K0+63] (diagonal rows; also the B operand) and Pq[:, R0..R0+63] k-major
        double2 va[8], vt[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int mm = idx >> 5, c = (idx & 31) * 2;
            va[it] = *reinterpret_cast<const double2*>(Pq + (int64_t)mm * ld + K0 + c);
            if (HAS_T)
                vt[it] = *reinterpret_cast<const double2*>(Pq + (int64_t)mm * ld + R0 + c);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int mm = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2*>(&Ads[mm * kLdsRow + c]) = va[it];
            if (HAS_T)
                *reinterpret_cast<double2*>(&Ats[mm * kLdsRow + c]) = vt[it];
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int row = (ks * 4 + fk) * kLdsRow;
            const double ad = -Ads[row + 16 * w + fr];
            const double at = HAS_T ? -Ats[row + 16 * w + fr] : 0.0;
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const double b = Ads[row + 16 * tj + fr];
                const double adm = (tj <= w) ? ad : 0.0;
                Dacc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(adm, b, Dacc[tj], 0, 0, 0);
                if (HAS_T)
                    Tacc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(at, b, Tacc[tj], 0, 0, 0);
            }
        }
    }
    STAMP(1);
    bool ok = true;
    panel_round<0, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<8, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<16, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<24, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<32, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<40, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<48, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    panel_round<56, HAS_T>(w, lane, Dacc, Tacc, Pd, Pt, RA, RA, invd, ok);
    __syncthreads();
    STAMP(2);
    if (!HAS_T) {
        // `ok` is meaningful in wave 0 only
        if (tid == 0 && !ok)
            ctl->lin_fail = 1;
        if (tid < 64)
            dinv[K0 + tid] = invd[tid];
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            if (c <= r)
                Ld[(int64_t)k * 4096 + r * 64 + c] = RA[c * kLdT + r];
        }
        return;
    }
    STAMP(4);
    // L_ik = the scaled columns collected in R: coalesced stores to S and, transposed, to P
    const double* R = RA;
    for (int idx = tid; idx < 64 * 32; idx += 256) {
        const int rr = idx >> 5, c = (idx & 31) * 2;
        if (R0 + rr <= n_pad)
            *reinterpret_cast<double2*>(S + (int64_t)(R0 + rr) * ld + K0 + c)
                = make_double2(R[rr * kLd + c], R[rr * kLd + c + 1]);
    }
    {
        const int rr = tid & 63;
        if (R0 + rr <= n_pad)
            for (int c = tid >> 6; c < 64; c += 4)
                P[(int64_t)c * ld + R0 + rr] = R[rr * kLd + c];
    }
    STAMP(5);
}

constexpr int kPanelSmem = 64 * kLdT + 4 * 64 * kPs + 64 + 2 * 64 * kLdsRow;   // doubles (kPs = 9: 4 x 576)
constexpr int kUpdateSmem = 4 * 64 * kLdsRow;   // two operand slices, double-buffered: exactly the 160 KB of a CU
constexpr int kStepSmem = kPanelSmem > kUpdateSmem ? kPanelSmem : kUpdateSmem;

__device__ __forceinline__ void chol_panel_wg(LmCtl* ctl, double* __restrict__ S, int ld, int n_pad, int k,
                                              double* __restrict__ P, const double* __restrict__ Pprev,
                                              const double* __restrict__ Pprev2, double* __restrict__ dinv,
                                              double* __restrict__ Ld, double* smem)
{
    double* RA = smem;                     // workgroup 0: L^T (stride kLdT); others: result tile R (stride kLd)
    double* Pd = RA + 64 * kLdT;
    double* Pt = Pd + 2 * 64 * kPs;
    double* invd = Pt + 2 * 64 * kPs;
    double* Ads = invd + 64;               // previous panel, diagonal rows (k-major); 16-byte aligned offsets
    double* Ats = Ads + 64 * kLdsRow;      // previous panel, this workgroup's rows
    if (blockIdx.x == 0)
        panel_body<false>(ctl, S, ld, n_pad, k, P, Pprev, Pprev2, dinv, Ld, RA, Pd, Pt, invd, Ads, Ats);
    else
        panel_body<true>(ctl, S, ld, n_pad, k, P, Pprev, Pprev2, dinv, Ld, RA, Pd, Pt, invd, Ads, Ats);
}

#ifdef VMM_STAMPS
extern "C" int vmm_ba_debug_read_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 64 ? n : 64));
}
#endif

// Trailing update of block column k: A_ij -= L_ik L_jk^T for k+1 < j <= i (the rhs row block included;
// block column k+1 is left to the next panel kernel, see the look-ahead note there)
// with K = 64 taken from the transposed panel P (64 x ld, row m = panel column m).  One workgroup per
// 64x64 tile; the whole K extent of both operands (2 x 32 KB) and the C tile are requested up front
// so the kernel pays one memory latency, then 16 k-steps of four v_mfma_f64_16x16x4_f64 per wave.
typedef double double2v __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ void update_tile_index(int n_blk, int k, int t, int& bi, int& bj)
{
    // tile index -> (bi, bj): columns k+2..min(bi, n_blk-1) (block column k+1 is updated lazily by the
    // panel of that column), rows k+2..n_blk.  Row q = bi - (k+2) holds q + 1 tiles, except the last row (the
    // right-hand side, bi = n_blk), which has as many as the row before it: closed form, no search (a search
    // from the first row costs ~50 cycles per row, 2 us at 94 rows -- as much as the tile's MFMAs).
    const int n_rows = n_blk - (k + 2) + 1;                 // rows k+2 .. n_blk
    const int before_last = (n_rows - 1) * n_rows / 2;      // tiles in front of the last row
    int q;
    if (t >= before_last) {
        q = n_rows - 1;
        t -= before_last;
    } else {
        q = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        // guard the rounding of the square root
        while (q * (q + 1) / 2 > t)
            --q;
        while ((q + 1) * (q + 2) / 2 <= t)
            ++q;
        t -= q * (q + 1) / 2;
    }
    bi = k + 2 + q;
    bj = k + 2 + t;
}

// The tiles of a launch's trailing update are handed out through a counter (one atomic per tile, fetched two tiles
// ahead of its use): the dedicated update workgroups start at once, the panel workgroups of the same launch join when
// their panel is stored -- at n = 6000 a panel takes ~26 us of a launch that lasts up to 150 us, and the 95 CUs of the
// panel workgroups used to idle for the rest of it.  With more tiles than compute units the operands of the NEXT tile
// are requested before the MFMAs of the current one and parked in the other half of the LDS, and the C tile is
// requested at the start of its own iteration and only added after the 16 k-steps: a tile costs its MFMAs plus one
// barrier instead of a full memory latency.  No register array lives across the loop back-edge (those end up in
// scratch).  The order in which workgroups take tiles does not touch the result: a tile is updated by exactly one.
__device__ __forceinline__ void chol_update_wg(double* __restrict__ S, int ld, int n_blk, int k, unsigned* counter,
                                               int n_tiles, const double* __restrict__ P, double* smem)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int fk = lane >> 4, fi = lane & 15;
    // four ints in the padding columns of the first LDS row (the operand tiles use columns 0..63 of every row)
    volatile int* slot = reinterpret_cast<volatile int*>(smem + 64);
    if (tid == 0) {
        slot[0] = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slot[1] = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int t = slot[0], tn = slot[1];
    const int u = t, n_wg = 0;   // (names the diagnostic stamps refer to)
    (void)u;
    (void)n_wg;
    if (t >= n_tiles)
        return;
    int bi, bj;
    update_tile_index(n_blk, k, t, bi, bj);
    {
        double2 va[8], vb[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            va[it] = *reinterpret_cast<const double2*>(P + (int64_t)m * ld + bi * kNB + c);
            vb[it] = *reinterpret_cast<const double2*>(P + (int64_t)m * ld + bj * kNB + c);   // diagonal tile: same lines
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2*>(&smem[m * kLdsRow + c]) = va[it];
            *reinterpret_cast<double2*>(&smem[(64 + m) * kLdsRow + c]) = vb[it];
        }
    }
    __syncthreads();
    int cur = 0;
    for (int iter = 0;; ++iter) {
        const double* As = smem + cur * 128 * kLdsRow;
        const double* Bs = As + 64 * kLdsRow;
        const int I0 = bi * kNB, J0 = bj * kNB;
        // the tile after the next one, read by everybody behind this iteration's closing barrier
        if (tid == 0)
            slot[2 + (iter & 1)] = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The requests of this tile's C values and of the NEXT tile's operands are issued as volatile asm:
        // written as plain loads, LLVM sinks them below the MFMA loop to their first use (measured: the
        // memory latency then adds to the MFMA time, 6.6 us per tile instead of ~3).  The results are only
        // touched after the matching s_waitcnt below, which takes them as read-write operands.
        USTAMP(40);
        double creg[2][2][4];
        const double* pc[16];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pc[8 * a + 4 * b + r] = S + (int64_t)(I0 + wi * 32 + a * 16 + fk + 4 * r) * ld + J0 + wj * 32 + b * 16 + fi;
        const bool more = tn < n_tiles;   // workgroup-uniform
        // The last tile re-requests itself (result unused).
        int nbi = bi, nbj = bj;
        if (more)
            update_tile_index(n_blk, k, tn, nbi, nbj);
        double2v va[8], vb[8];
        const double* pa[8];
        const double* pb[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            pa[it] = P + (int64_t)m * ld + nbi * kNB + c;
            pb[it] = P + (int64_t)m * ld + nbj * kNB + c;
        }
        USTAMP(41);
        double4_t acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
        // 16 k-steps; the LDS operands of step ks+1 are read before the MFMAs of step ks, and one of the 16
        // operand requests of the next tile is issued per step (VMEM issue slots beside the MFMAs)
        double a0 = -As[fk * kLdsRow + wi * 32 + fi], a1 = -As[fk * kLdsRow + wi * 32 + 16 + fi];
        double b0 = Bs[fk * kLdsRow + wj * 32 + fi], b1 = Bs[fk * kLdsRow + wj * 32 + 16 + fi];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            // one pair of requests per k-step (issuing all 32 in the first four steps was measured slower: 4.02 against
            // 3.69 ms per factorisation at n = 6000 -- the loop is bound by the memory system's throughput, not by latency)
            if (ks < 8) {   // this tile's C values (HBM, the longer latency) first ...
                __asm__ volatile("global_load_dwordx2 %0, %1, off nt"
                                 : "=&v"(creg[(2 * ks) >> 3][((2 * ks) >> 2) & 1][(2 * ks) & 3]) : "v"(pc[2 * ks]) : "memory");
                __asm__ volatile("global_load_dwordx2 %0, %1, off nt"
                                 : "=&v"(creg[(2 * ks + 1) >> 3][((2 * ks + 1) >> 2) & 1][(2 * ks + 1) & 3]) : "v"(pc[2 * ks + 1]) : "memory");
            } else {        // ... then the next tile's operands (L2)
                __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(va[ks - 8]) : "v"(pa[ks - 8]) : "memory");
                __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(vb[ks - 8]) : "v"(pb[ks - 8]) : "memory");
            }
            double na0 = 0.0, na1 = 0.0, nb0 = 0.0, nb1 = 0.0;
            if (ks < 15) {
                const int row = ((ks + 1) * 4 + fk) * kLdsRow;
                na0 = -As[row + wi * 32 + fi];
                na1 = -As[row + wi * 32 + 16 + fi];
                nb0 = Bs[row + wj * 32 + fi];
                nb1 = Bs[row + wj * 32 + 16 + fi];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            a0 = na0;
            a1 = na1;
            b0 = nb0;
            b1 = nb1;
        }
#ifdef VMM_STAMPS
        __asm__ volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][1][3]) : "memory");
#endif
        USTAMP(42);
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(creg[0][0][0]), "+v"(creg[0][0][1]), "+v"(creg[0][0][2]), "+v"(creg[0][0][3]),
                           "+v"(creg[0][1][0]), "+v"(creg[0][1][1]), "+v"(creg[0][1][2]), "+v"(creg[0][1][3]),
                           "+v"(creg[1][0][0]), "+v"(creg[1][0][1]), "+v"(creg[1][0][2]), "+v"(creg[1][0][3])
                         :
                         : "memory");
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(creg[1][1][0]), "+v"(creg[1][1][1]), "+v"(creg[1][1][2]), "+v"(creg[1][1][3]),
                           "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]), "+v"(va[5]), "+v"(va[6]),
                           "+v"(va[7])
                         :
                         : "memory");
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]), "+v"(vb[4]), "+v"(vb[5]), "+v"(vb[6]),
                           "+v"(vb[7])
                         :
                         : "memory");
        USTAMP(43);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    // non-temporal, like the loads of these values: a tile of C is touched once per launch, and kept out
                    // of the L2 it leaves the transposed panel (3 MB, read by every workgroup for every tile) resident --
                    // the update is bound by memory traffic (16 B of C + 16 B of operands per 128 flops), not by the MFMAs
                    __builtin_nontemporal_store(creg[a][b][r] + acc[a][b][r],
                                                &S[(int64_t)(I0 + wi * 32 + a * 16 + fk + 4 * r) * ld + J0 + wj * 32 + b * 16 + fi]);
        USTAMP(44);
        // park the next tile's operands in the other half (nobody reads it during this iteration)
        double* An = smem + (cur ^ 1) * 128 * kLdsRow;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2v*>(&An[m * kLdsRow + c]) = va[it];
            *reinterpret_cast<double2v*>(&An[(64 + m) * kLdsRow + c]) = vb[it];
        }
        USTAMP(45);
        __syncthreads();
        USTAMP(46);
        cur ^= 1;
        bi = nbi;
        bj = nbj;
        t = tn;
        tn = slot[2 + (iter & 1)];
        if (t >= n_tiles)
            break;
    }
}

// Rank-128 trailing update: the transposed panels PA (block column c0-3) and PB (c0-2) applied in ONE visit of each C tile
// of the block columns >= c0 (rows >= column, the right-hand side row included).  A rank-64 visit moves 16 B of C per
// 128 flops and the launch is bound by that traffic (measured at n = 6000: the MFMA work of two updates in one visit
// costs 1.35x one visit, not 2x); the pair halves it.  Tile t of the pair's list: first block column c0 (needed by the
// next panel), then the triangle of the columns > c0 in update_tile_index order; the list is worked off by two
// consecutive launches (tiles [t0, t1) each, handed out by `counter` as in chol_update_wg).
// One loop iteration = one tile = two halves of 16 k-steps: half 0 multiplies the PA operands (parked in LDS half `0`)
// while the tile's C values (non-temporal) and its PB operands are requested, half 1 multiplies the PB operands
// (LDS half `1`) while the NEXT tile's PA operands are requested; C is added and stored behind half 1.  As in
// chol_update_wg the requests are volatile asm, touched only behind the matching s_waitcnt (tools/check_chol_asm.py).
__host__ __device__ __forceinline__ void pair_tile_index(int n_blk, int c0, int t, int& bi, int& bj)
{
    const int n_first = n_blk - c0 + 1;   // block column c0: rows c0 .. n_blk
    if (t < n_first) {
        bi = c0 + t;
        bj = c0;
    } else {
        update_tile_index(n_blk, c0 - 1, t - n_first, bi, bj);
    }
}

__device__ __forceinline__ void chol_update2_wg(double* __restrict__ S, int ld, int n_blk, int c0, unsigned* counter,
                                                int t0, int t1, const double* __restrict__ PA,
                                                const double* __restrict__ PB, double* smem)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int fk = lane >> 4, fi = lane & 15;
    volatile int* slot = reinterpret_cast<volatile int*>(smem + 64);   // padding columns of the first LDS row
    if (tid == 0) {
        slot[0] = t0 + (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slot[1] = t0 + (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int t = slot[0], tn = slot[1];
    if (t >= t1)
        return;
    int bi, bj;
    pair_tile_index(n_blk, c0, t, bi, bj);
    double* const L0 = smem;                    // PA operands: A rows 0..63, B rows 64..127 (k-major)
    double* const L1 = smem + 128 * kLdsRow;    // PB operands
    {
        double2 va[8], vb[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            va[it] = *reinterpret_cast<const double2*>(PA + (int64_t)m * ld + bi * kNB + c);
            vb[it] = *reinterpret_cast<const double2*>(PA + (int64_t)m * ld + bj * kNB + c);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2*>(&L0[m * kLdsRow + c]) = va[it];
            *reinterpret_cast<double2*>(&L0[(64 + m) * kLdsRow + c]) = vb[it];
        }
    }
    __syncthreads();
    for (int iter = 0;; ++iter) {
        const int I0 = bi * kNB, J0 = bj * kNB;
        if (tid == 0)   // the tile after the next one, read by everybody behind this iteration's closing barrier
            slot[2 + (iter & 1)] = t0 + (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double creg[2][2][4];
        double2v va[8], vb[8];
        const double* pa[8];
        const double* pb[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            pa[it] = PB + (int64_t)m * ld + bi * kNB + c;
            pb[it] = PB + (int64_t)m * ld + bj * kNB + c;
        }
        double4_t acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
        // ---- half 0: PA operands; requests: this tile's PB operands, then its C values.  The C values (HBM, the longer
        // latency) are only needed behind half 1: the wait at the end of this half leaves the 16 most recent requests --
        // exactly them -- in flight (loads return in order), so they have both halves to arrive
        {
            // (the addresses of the C values only live in this half: they are recomputed for the stores -- kept across
            // half 1 their 32 registers push the compiler into copying `creg` while its loads are still in flight)
            const double* pc[16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pc[8 * a + 4 * b + r] = S + (int64_t)(I0 + wi * 32 + a * 16 + fk + 4 * r) * ld + J0 + wj * 32 + b * 16 + fi;
            const double* As = L0;
            const double* Bs = L0 + 64 * kLdsRow;
            double a0 = -As[fk * kLdsRow + wi * 32 + fi], a1 = -As[fk * kLdsRow + wi * 32 + 16 + fi];
            double b0 = Bs[fk * kLdsRow + wj * 32 + fi], b1 = Bs[fk * kLdsRow + wj * 32 + 16 + fi];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks < 8) {
                    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(va[ks]) : "v"(pa[ks]) : "memory");
                    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(vb[ks]) : "v"(pb[ks]) : "memory");
                } else {
                    const int q = 2 * (ks - 8);
                    __asm__ volatile("global_load_dwordx2 %0, %1, off nt"
                                     : "=&v"(creg[q >> 3][(q >> 2) & 1][q & 3]) : "v"(pc[q]) : "memory");
                    __asm__ volatile("global_load_dwordx2 %0, %1, off nt"
                                     : "=&v"(creg[(q + 1) >> 3][((q + 1) >> 2) & 1][(q + 1) & 3]) : "v"(pc[q + 1]) : "memory");
                }
                double na0 = 0.0, na1 = 0.0, nb0 = 0.0, nb1 = 0.0;
                if (ks < 15) {
                    const int row = ((ks + 1) * 4 + fk) * kLdsRow;
                    na0 = -As[row + wi * 32 + fi];
                    na1 = -As[row + wi * 32 + 16 + fi];
                    nb0 = Bs[row + wj * 32 + fi];
                    nb1 = Bs[row + wj * 32 + 16 + fi];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0;
                a1 = na1;
                b0 = nb0;
                b1 = nb1;
            }
        }
        __asm__ volatile("s_waitcnt vmcnt(16)"
                         : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]), "+v"(va[5]), "+v"(va[6]),
                           "+v"(va[7])
                         :
                         : "memory");
        __asm__ volatile("s_waitcnt vmcnt(16)"
                         : "+v"(vb[0]), "+v"(vb[1]), "+v"(vb[2]), "+v"(vb[3]), "+v"(vb[4]), "+v"(vb[5]), "+v"(vb[6]),
                           "+v"(vb[7])
                         :
                         : "memory");
        // park the PB operands in the other half (its last readers finished before the previous closing barrier)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2v*>(&L1[m * kLdsRow + c]) = va[it];
            *reinterpret_cast<double2v*>(&L1[(64 + m) * kLdsRow + c]) = vb[it];
        }
        __syncthreads();
        // ---- half 1: PB operands; requests: the NEXT tile's PA operands (the last tile re-requests itself, unused)
        const bool more = tn < t1;   // workgroup-uniform
        int nbi = bi, nbj = bj;
        if (more)
            pair_tile_index(n_blk, c0, tn, nbi, nbj);
        double2v wa[8], wb[8];
        const double* qa[8];
        const double* qb[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            qa[it] = PA + (int64_t)m * ld + nbi * kNB + c;
            qb[it] = PA + (int64_t)m * ld + nbj * kNB + c;
        }
        {
            const double* As = L1;
            const double* Bs = L1 + 64 * kLdsRow;
            double a0 = -As[fk * kLdsRow + wi * 32 + fi], a1 = -As[fk * kLdsRow + wi * 32 + 16 + fi];
            double b0 = Bs[fk * kLdsRow + wj * 32 + fi], b1 = Bs[fk * kLdsRow + wj * 32 + 16 + fi];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks < 8) {
                    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wa[ks]) : "v"(qa[ks]) : "memory");
                    __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wb[ks]) : "v"(qb[ks]) : "memory");
                }
                double na0 = 0.0, na1 = 0.0, nb0 = 0.0, nb1 = 0.0;
                if (ks < 15) {
                    const int row = ((ks + 1) * 4 + fk) * kLdsRow;
                    na0 = -As[row + wi * 32 + fi];
                    na1 = -As[row + wi * 32 + 16 + fi];
                    nb0 = Bs[row + wj * 32 + fi];
                    nb1 = Bs[row + wj * 32 + 16 + fi];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0;
                a1 = na1;
                b0 = nb0;
                b1 = nb1;
            }
        }
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(creg[0][0][0]), "+v"(creg[0][0][1]), "+v"(creg[0][0][2]), "+v"(creg[0][0][3]),
                           "+v"(creg[0][1][0]), "+v"(creg[0][1][1]), "+v"(creg[0][1][2]), "+v"(creg[0][1][3]),
                           "+v"(creg[1][0][0]), "+v"(creg[1][0][1]), "+v"(creg[1][0][2]), "+v"(creg[1][0][3])
                         :
                         : "memory");
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(creg[1][1][0]), "+v"(creg[1][1][1]), "+v"(creg[1][1][2]), "+v"(creg[1][1][3]),
                           "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(wa[4]), "+v"(wa[5]), "+v"(wa[6]),
                           "+v"(wa[7])
                         :
                         : "memory");
        __asm__ volatile("s_waitcnt vmcnt(0)"
                         : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]), "+v"(wb[4]), "+v"(wb[5]), "+v"(wb[6]),
                           "+v"(wb[7])
                         :
                         : "memory");
        // (stores behind the wait: on gfx9 they count in vmcnt too)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __builtin_nontemporal_store(creg[a][b][r] + acc[a][b][r],
                                                &S[(int64_t)(I0 + wi * 32 + a * 16 + fk + 4 * r) * ld + J0 + wj * 32 + b * 16 + fi]);
        // park the next tile's PA operands (half 0 was last read before the barrier in the middle of this iteration)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tid;
            const int m = idx >> 5, c = (idx & 31) * 2;
            *reinterpret_cast<double2v*>(&L0[m * kLdsRow + c]) = wa[it];
            *reinterpret_cast<double2v*>(&L0[(64 + m) * kLdsRow + c]) = wb[it];
        }
        __syncthreads();
        bi = nbi;
        bj = nbj;
        t = tn;
        tn = slot[2 + (iter & 1)];
        if (t >= t1)
            break;
    }
}

// One step of L^T y = w (w lives in row n_pad of S).  Launched for kb = n_blk-1 .. 0 with kb+1
// workgroups of 256 threads: workgroup m first applies y_{kb+1} to w_m (64x64 transposed GEMV split
// over the four waves), then workgroup kb solves its diagonal block four unknowns per round.
__global__ __launch_bounds__(256) void k_backsolve_step(const LmCtl* ctl, double* __restrict__ S, int ld,
                                                        int n_pad, int n_blk, int kb, double* __restrict__ y,
                                                        const double* __restrict__ dinv,
                                                        const double* __restrict__ Ld)
{
    if (ctl->done || ctl->lin_fail)
        return;
    __shared__ double red[4][64];
    __shared__ double L[64 * kLd];
    __shared__ double ws[64];
    __shared__ double di[64];
    const int m = blockIdx.x;
    const int tid = threadIdx.x;
    const int c = tid & 63, part = tid >> 6;
    double* w = S + (int64_t)n_pad * ld;
    double wc = 0.0;
    if (kb + 1 < n_blk) {
        const int R0 = (kb + 1) * kNB + part * 16;
        const double* Lb = S + (int64_t)R0 * ld + m * kNB + c;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc += Lb[(int64_t)r * ld] * y[R0 + r];
        red[part][c] = acc;
    }
    if (m == kb) {
        // stage the diagonal block while the partial sums settle
        const int K0 = kb * kNB;
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, cc = idx & 63;
            L[r * kLd + cc] = (cc <= r) ? Ld[(int64_t)kb * 4096 + r * 64 + cc] : 0.0;
        }
        if (tid < 64)
            di[tid] = dinv[K0 + tid];
    }
    __syncthreads();
    if (part == 0) {
        wc = w[m * kNB + c];
        if (kb + 1 < n_blk) {
            wc -= (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            w[m * kNB + c] = wc;
        }
    }
    if (m != kb)
        return;
    // single wave from here on (part == 0); other waves only keep the barriers company
    double yj = 0.0;
    for (int j0 = 60; j0 >= 0; j0 -= 4) {
        if (part == 0)
            ws[c] = wc;
        __syncthreads();
        if (part == 0) {
            // L4^T v = w4 with L4 the lower 4x4 pivot block at (j0, j0)
            const double* D = L + j0 * kLd + j0;
            const double v3 = ws[j0 + 3] * di[j0 + 3];
            const double v2 = (ws[j0 + 2] - D[3 * kLd + 2] * v3) * di[j0 + 2];
            const double v1 = (ws[j0 + 1] - D[2 * kLd + 1] * v2 - D[3 * kLd + 1] * v3) * di[j0 + 1];
            const double v0 = (ws[j0] - D[kLd] * v1 - D[2 * kLd] * v2 - D[3 * kLd] * v3) * di[j0];
            if (c >= j0 && c < j0 + 4)
                yj = (c == j0) ? v0 : (c == j0 + 1 ? v1 : (c == j0 + 2 ? v2 : v3));
            if (c < j0)
                wc -= L[j0 * kLd + c] * v0 + L[(j0 + 1) * kLd + c] * v1 + L[(j0 + 2) * kLd + c] * v2
                    + L[(j0 + 3) * kLd + c] * v3;
        }
        __syncthreads();
    }
    if (part == 0)
        y[kb * kNB + c] = yj;
}

// Whole back-substitution L^T y = w in ONE launch: workgroup p owns block m = n_blk-1-p and depends on
// the workgroups before it (dispatched earlier), which publish their 64 unknowns as self-validating
// granules (cdna_hip_programming.md Guideline 16, R2: "the data IS the flag"): every double travels as two
// 8-byte {tag = epoch, 32 value bits} words written by ONE aligned agent-scope (sc1) store each; one wave
// of the consumer re-reads its 128 granules with sc1 loads until every tag carries this solve's epoch and
// hands the values to the other waves through LDS.  One L2 round trip per hop instead of two (flag, then
// payload), no drain + flag store on the producer side; the kernel boundaries of the per-block version disappear.
//
// What a hop costs (tools/gpu_chain_stamps.sh): ~0.45 us from a block's publication to its successor seeing it, and
// -- before this form -- ~0.95 us of work behind it: the product with L(m+1,m)^T, a reduction over the four waves,
// the product with the inverse of the diagonal block, another reduction (four barriers).  Only ONE product has to
// wait for y_{m+1}:
//     y_m = Linv_m^T (w_m - sum_{j>m+1} L(j,m)^T y_j)  -  (L(m+1,m) Linv_m)^T y_{m+1}  =  u_m - B_m^T y_{m+1},
// u_m is finished one hop earlier and B_m (a 64x64x64 product) while the workgroup waits for the chain to reach it.
// (Two blocks per workgroup, 10 hops instead of 19, was built first and changed nothing: the work, not the hand-off,
// was the larger part of a hop.)
// Every spin is bounded: a workgroup that gives up raises LmCtl::sync_timeout (NOT lin_fail: a stalled workgroup is
// not an indefinite matrix) and pauses the loop (done = 2); it still publishes, so that no other workgroup is left
// waiting.  The host then redoes this pass's factorisation on the path without inter-workgroup waits.
constexpr unsigned kSpinLimit = 1u << 22;

// one thread: this pass gave up waiting in kernel `bit` (1 dataflow factorisation, 2 back-substitution chain)
__device__ __forceinline__ void raise_sync_timeout(LmCtl* ctl, int bit)
{
    atomicOr(&ctl->sync_timeout, bit);
    atomicExch(&ctl->done, 2);
}

int backsolve_chain_workgroups(int n_blk) { return n_blk; }

// The workgroup that finishes LAST retires the epoch (every workgroup has read the old value at its start by then; with
// a tree ordering block 0 is not the last to finish any more) and leaves the counter at zero for the next launch.
// (dense chains end with block 0 by construction: it retires the epoch with a plain store, as before -- the returning
// atomic costs the last workgroup, i.e. the launch, ~1-2 us)
__device__ __forceinline__ void chain_block_done(unsigned* n_done, int n_blk, unsigned* epoch_word, unsigned epoch,
                                                 bool tree, int m)
{
    if (!tree) {
        if (m == 0)
            *epoch_word = epoch;
        return;
    }
    if (atomicAdd(n_done, 1u) == (unsigned)n_blk - 1u) {
        *n_done = 0u;
        *epoch_word = epoch;
    }
}

#ifdef VMM_STAMPS
__device__ unsigned long long g_chain_stamps[128][4];   // [block]: start, last dependency seen, published
#define CH_RT(blk, slot)                                                                  \
    do {                                                                                  \
        if (threadIdx.x == 0)                                                             \
            g_chain_stamps[(blk) & 127][slot] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)
extern "C" int vmm_ba_debug_read_chain_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_stamps), sizeof(unsigned long long) * 128 * 4);
}
#else
#define CH_RT(blk, slot)
#endif

__global__ __launch_bounds__(256) void k_backsolve_chain(LmCtl* ctl, const double* __restrict__ S, int ld,
                                                         int n_pad, int n_blk, double* y,
                                                         const double* __restrict__ dinv, unsigned long long* gran,
                                                         unsigned* epoch_word, const double* __restrict__ Ld,
                                                         const double* __restrict__ Linv)
{
    // (a give-up inside the dataflow factorisation before this launch has set done = 2: every workgroup that stops
    // waiting raises it itself, report_give_up)
    // (the abort word itself is not looked at here: the epoch it is compared with is being retired by this very launch, and
    // every workgroup of the factorisation that gave up has raised done = 2 itself before that kernel ended)
    if (ctl->done || ctl->lin_fail) {
        // the factorisation before this launch may have tagged granules with the current epoch: retire it even
        // when the solve is skipped (every workgroup of this launch leaves here, so nobody needs the old value)
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
            *epoch_word = *epoch_word + 1u;
        return;
    }
    const unsigned spin_limit = (ctl->spin_limit_chain && (ctl->spin_wg < 0 || ctl->spin_wg == (int)blockIdx.x))
                                    ? ctl->spin_limit_chain : kSpinLimit;
    __shared__ double L[64 * kLd];    // L(m+1, m) for the product B_m
    __shared__ double Li[64 * kLd];   // Linv_m
    __shared__ double red[4][64];
    __shared__ double ws[64];
    __shared__ double ys[2][64];
    __shared__ double sB[16][256];    // B_m: [row within a wave's 16][thread that owns the column]
    __shared__ int s_timeout;
    const int m = n_blk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x;
    const int c = tid & 63, part = tid >> 6;
    const unsigned epoch = *epoch_word + 1u;   // every workgroup reads it before workgroup n_blk-1 (the last) bumps it
    const int K0 = m * kNB;
    if (tid == 0)
        s_timeout = 0;
    CH_RT(m, 0);
    // wave 0 sweeps block j's granules into ysj: lane c owns unknown c (two granules)
    auto receive = [&](const int j, double* ysj) {
        if (part == 0) {
            const unsigned long long* g = gran + 2 * (int64_t)(j * kNB + c);
            unsigned long long x0, x1;
            for (unsigned n = 0;;) {
                x0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
                if (__all(ok) && spin_limit != 1u)   // a limit of 1 (debugging) gives up even on valid data
                    break;
                __builtin_amdgcn_s_sleep(1);
                if (++n >= spin_limit) {   // wave-uniform give-up: reported as a synchronisation time-out below
                    s_timeout = 1;
                    break;
                }
            }
            ysj[c] = __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
        }
    };
    auto publish = [&](const double yv) {   // wave 0
        const unsigned long long bits = (unsigned long long)__double_as_longlong(yv);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        unsigned long long* g = gran + 2 * (int64_t)(K0 + c);
        __hip_atomic_store(g, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[K0 + c] = yv;   // for the kernels after this launch
    };
    if (m == n_blk - 1) {
        // The last block is the first in the chain and has no inverse (nothing runs beside its factorisation).  ONE
        // wave solves L^T y = w by columns, lane c holding w[c]: 64 steps of (broadcast y_j from lane j, one
        // multiply-add per lane) on registers only -- no barrier, no LDS in the dependent chain.
        if (part == 0) {
            double lcol[64];
#pragma unroll
            for (int jj = 0; jj < 64; ++jj)
                lcol[jj] = (c <= jj) ? Ld[(int64_t)m * 4096 + jj * 64 + c] : 0.0;
            const double dic = dinv[K0 + c];
            double wv = S[(int64_t)n_pad * ld + K0 + c];
            double yv = 0.0;
#pragma unroll
            for (int jj = 63; jj >= 0; --jj) {
                // v_readlane (jj is a constant), not a cross-lane permute through the LDS
                const long long wb = __double_as_longlong(wv * dic);
                const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)wb, jj);
                const unsigned w1 = (unsigned)__builtin_amdgcn_readlane((int)(wb >> 32), jj);
                const double yj = __longlong_as_double((long long)(((unsigned long long)w1 << 32) | w0));
                yv = (c == jj) ? yj : yv;
                wv = (c < jj) ? wv - lcol[jj] * yj : wv;
            }
            publish(yv);
        }
        CH_RT(m, 2);
        if (tid == 0 && m == 0)
            *epoch_word = epoch;   // a single block: also the end of the chain
        return;   // no wait, so no timeout
    }
    // ---- B_m = L(m+1, m) Linv_m on the matrix cores: wave `part` computes rows part*16 .. part*16+15 ----
    // (as 16 x 64 dot products per thread with broadcast LDS reads it took 15 us: every workgroup was late for its hop)
    double li[16];
    {
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, cc = idx & 63;
            L[r * kLd + cc] = S[(int64_t)((m + 1) * kNB + r) * ld + K0 + cc];
            Li[r * kLd + cc] = Linv[(int64_t)m * 4096 + r * 64 + cc];   // lower triangular, zero above the diagonal
        }
        __syncthreads();
        // li: my 16 rows of column c of Linv_m, for u_m = Linv_m^T t
#pragma unroll
        for (int r = 0; r < 16; ++r)
            li[r] = Li[(part * 16 + r) * kLd + c];
        // v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
        // C[row = (lane >> 4) + 4 reg][col = lane & 15]
        const int fi = c & 15, fk = c >> 4;
        double4_t accB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            accB[t] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double av = L[(part * 16 + fi) * kLd + 4 * ks + fk];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                accB[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Li[(4 * ks + fk) * kLd + 16 * t + fi], accB[t], 0, 0, 0);
        }
        // to the layout the hop reads: sB[row within my 16][workgroup thread that owns the column]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sB[fk + 4 * r][part * 64 + 16 * t + fi] = accB[t][r];
        __syncthreads();
    }
    // ---- the blocks behind m+1: acc = sum_j L(j, m)^T y_j ----
    double acc = 0.0;
    double lt[16], ln[16];
    int j = n_blk - 1;
    if (j > m + 1) {
        const double* Lb = S + (int64_t)(j * kNB + part * 16) * ld + K0 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            lt[r] = Lb[(int64_t)r * ld];
    }
    for (; j > m + 1; --j) {
        if (j - 1 > m + 1) {   // next tile requested before the wait
            const double* Lb = S + (int64_t)((j - 1) * kNB + part * 16) * ld + K0 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ln[r] = Lb[(int64_t)r * ld];
        }
        double* ysj = ys[j & 1];
        receive(j, ysj);
        __syncthreads();   // also orders the reuse of ys[j & 1] two hops later
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc += lt[r] * ysj[part * 16 + r];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            lt[r] = ln[r];
    }
    // ---- u_m = Linv_m^T (w_m - acc): one hop ahead of the value it will be combined with ----
    red[part][c] = acc;
    __syncthreads();
    if (part == 0)
        ws[c] = S[(int64_t)n_pad * ld + K0 + c] - ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    __syncthreads();
    double a2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        a2 += li[r] * ws[part * 16 + r];
    __syncthreads();   // everyone has read red[] above
    red[part][c] = a2;
    __syncthreads();
    double u = 0.0;
    if (part == 0)
        u = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    // ---- the hop: y_m = u_m - B_m^T y_{m+1} ----
    double* ysj = ys[(m + 1) & 1];
    receive(m + 1, ysj);
    __syncthreads();   // also: everyone has read red[] above
    CH_RT(m, 1);
    double a3 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        a3 += sB[r][tid] * ysj[part * 16 + r];
    red[part][c] = a3;
    __syncthreads();
    if (part == 0)
        publish(u - ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])));
    CH_RT(m, 2);
    if (tid == 0) {
        if (s_timeout)
            raise_sync_timeout(ctl, 2);
        if (m == 0)
            *epoch_word = epoch;   // block 0 is the end of the chain: every other workgroup has read the old value
    }
}

// The same chain for a factor with a block structure (tree orderings of the kept family, `nz` as in DfArgs): block m waits
// only for the blocks j > m with L(j, m) != 0, its hop is on the nearest of them (the parent in the elimination tree)
// instead of m + 1, and the workgroup that finishes LAST retires the epoch (block 0 is no longer the last).  A kernel of
// its own: the dense chain above stays exactly the code that was tuned (measured A/B on one box: the merged form cost
// the dense chain ~2 us).
__global__ __launch_bounds__(256) void k_backsolve_chain_tree(LmCtl* ctl, const double* __restrict__ S, int ld,
                                                         int n_pad, int n_blk, double* y,
                                                         const double* __restrict__ dinv, unsigned long long* gran,
                                                         unsigned* epoch_word, const double* __restrict__ Ld,
                                                         const double* __restrict__ Linv,
                                                         const unsigned long long* __restrict__ nz, unsigned* n_done)
{
    // (a give-up inside the dataflow factorisation before this launch has set done = 2: every workgroup that stops
    // waiting raises it itself, report_give_up)
    // (the abort word itself is not looked at here: the epoch it is compared with is being retired by this very launch, and
    // every workgroup of the factorisation that gave up has raised done = 2 itself before that kernel ended)
    if (ctl->done || ctl->lin_fail) {
        // the factorisation before this launch may have tagged granules with the current epoch: retire it even
        // when the solve is skipped (every workgroup of this launch leaves here, so nobody needs the old value)
        if (threadIdx.x == 0)
            chain_block_done(n_done, n_blk, epoch_word, *epoch_word + 1u, nz != nullptr, n_blk - 1 - (int)blockIdx.x);
        return;
    }
    const unsigned spin_limit = (ctl->spin_limit_chain && (ctl->spin_wg < 0 || ctl->spin_wg == (int)blockIdx.x))
                                    ? ctl->spin_limit_chain : kSpinLimit;
    __shared__ double L[64 * kLd];    // L(m+1, m) for the product B_m
    __shared__ double Li[64 * kLd];   // Linv_m
    __shared__ double red[4][64];
    __shared__ double ws[64];
    __shared__ double ys[2][64];
    __shared__ double sB[16][256];    // B_m: [row within a wave's 16][thread that owns the column]
    __shared__ int s_timeout;
    const int m = n_blk - 1 - (int)blockIdx.x;
    const int tid = threadIdx.x;
    const int c = tid & 63, part = tid >> 6;
    const unsigned epoch = *epoch_word + 1u;   // every workgroup reads it before workgroup n_blk-1 (the last) bumps it
    const int K0 = m * kNB;
    if (tid == 0)
        s_timeout = 0;
    CH_RT(m, 0);
    // wave 0 sweeps block j's granules into ysj: lane c owns unknown c (two granules)
    auto receive = [&](const int j, double* ysj) {
        if (part == 0) {
            const unsigned long long* g = gran + 2 * (int64_t)(j * kNB + c);
            unsigned long long x0, x1;
            for (unsigned n = 0;;) {
                x0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch;
                if (__all(ok) && spin_limit != 1u)   // a limit of 1 (debugging) gives up even on valid data
                    break;
                __builtin_amdgcn_s_sleep(1);
                if (++n >= spin_limit) {   // wave-uniform give-up: reported as a synchronisation time-out below
                    s_timeout = 1;
                    break;
                }
            }
            ysj[c] = __longlong_as_double((long long)(((x1 & 0xffffffffull) << 32) | (x0 & 0xffffffffull)));
        }
    };
    auto publish = [&](const double yv) {   // wave 0
        const unsigned long long bits = (unsigned long long)__double_as_longlong(yv);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        unsigned long long* g = gran + 2 * (int64_t)(K0 + c);
        __hip_atomic_store(g, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[K0 + c] = yv;   // for the kernels after this launch
    };
    if (m == n_blk - 1) {
        // The last block is the first in the chain and has no inverse (nothing runs beside its factorisation).  ONE
        // wave solves L^T y = w by columns, lane c holding w[c]: 64 steps of (broadcast y_j from lane j, one
        // multiply-add per lane) on registers only -- no barrier, no LDS in the dependent chain.
        if (part == 0) {
            double lcol[64];
#pragma unroll
            for (int jj = 0; jj < 64; ++jj)
                lcol[jj] = (c <= jj) ? Ld[(int64_t)m * 4096 + jj * 64 + c] : 0.0;
            const double dic = dinv[K0 + c];
            double wv = S[(int64_t)n_pad * ld + K0 + c];
            double yv = 0.0;
#pragma unroll
            for (int jj = 63; jj >= 0; --jj) {
                // v_readlane (jj is a constant), not a cross-lane permute through the LDS
                const long long wb = __double_as_longlong(wv * dic);
                const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)wb, jj);
                const unsigned w1 = (unsigned)__builtin_amdgcn_readlane((int)(wb >> 32), jj);
                const double yj = __longlong_as_double((long long)(((unsigned long long)w1 << 32) | w0));
                yv = (c == jj) ? yj : yv;
                wv = (c < jj) ? wv - lcol[jj] * yj : wv;
            }
            publish(yv);
        }
        CH_RT(m, 2);
        if (tid == 0)
            chain_block_done(n_done, n_blk, epoch_word, epoch, nz != nullptr, m);
        return;   // no wait, so no timeout
    }
    // The blocks of column m below the diagonal.  Dense: all of m+1 .. n_blk-1, and the hop waits for y_{m+1}.  With the
    // factor's block structure (tree ordering): only those with L(j, m) != 0; the nearest one, jp, is the block whose
    // unknowns arrive last (the parent in the elimination tree) and takes the place of m+1; without any, y_m = u_m.
    auto below = [&](int jj) { return !nz || nz_bit(nz, jj, m); };   // L(jj, m) may be non-zero
    int jp = m + 1;
    while (jp < n_blk && !below(jp))
        ++jp;
    const bool has_parent = jp < n_blk;
    // ---- B_m = L(jp, m) Linv_m on the matrix cores: wave `part` computes rows part*16 .. part*16+15 ----
    // (as 16 x 64 dot products per thread with broadcast LDS reads it took 15 us: every workgroup was late for its hop)
    double li[16];
    {
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, cc = idx & 63;
            L[r * kLd + cc] = has_parent ? S[(int64_t)(jp * kNB + r) * ld + K0 + cc] : 0.0;
            Li[r * kLd + cc] = Linv[(int64_t)m * 4096 + r * 64 + cc];   // lower triangular, zero above the diagonal
        }
        __syncthreads();
        // li: my 16 rows of column c of Linv_m, for u_m = Linv_m^T t
#pragma unroll
        for (int r = 0; r < 16; ++r)
            li[r] = Li[(part * 16 + r) * kLd + c];
        // v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
        // C[row = (lane >> 4) + 4 reg][col = lane & 15]
        const int fi = c & 15, fk = c >> 4;
        double4_t accB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            accB[t] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double av = L[(part * 16 + fi) * kLd + 4 * ks + fk];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                accB[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Li[(4 * ks + fk) * kLd + 16 * t + fi], accB[t], 0, 0, 0);
        }
        // to the layout the hop reads: sB[row within my 16][workgroup thread that owns the column]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sB[fk + 4 * r][part * 64 + 16 * t + fi] = accB[t][r];
        __syncthreads();
    }
    // ---- the blocks behind jp: acc = sum_j L(j, m)^T y_j, highest block first ----
    double acc = 0.0;
    double lt[16], ln[16];
    auto next_down = [&](int from) {   // the highest block below `from` (exclusive) and above jp with an entry; jp: none
        int jj = from - 1;
        while (jj > jp && !below(jj))
            --jj;
        return jj;
    };
    int j = has_parent ? next_down(n_blk) : jp;
    if (j > jp) {
        const double* Lb = S + (int64_t)(j * kNB + part * 16) * ld + K0 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            lt[r] = Lb[(int64_t)r * ld];
    }
    for (int parity = 0; j > jp; parity ^= 1) {
        const int jn = next_down(j);
        if (jn > jp) {   // next tile requested before the wait
            const double* Lb = S + (int64_t)(jn * kNB + part * 16) * ld + K0 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ln[r] = Lb[(int64_t)r * ld];
        }
        double* ysj = ys[parity];
        receive(j, ysj);
        __syncthreads();   // also orders the reuse of ys[parity] two blocks later
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc += lt[r] * ysj[part * 16 + r];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            lt[r] = ln[r];
        j = jn;
    }
    // ---- u_m = Linv_m^T (w_m - acc): one hop ahead of the value it will be combined with ----
    red[part][c] = acc;
    __syncthreads();
    if (part == 0)
        ws[c] = S[(int64_t)n_pad * ld + K0 + c] - ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    __syncthreads();
    double a2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        a2 += li[r] * ws[part * 16 + r];
    __syncthreads();   // everyone has read red[] above
    red[part][c] = a2;
    __syncthreads();
    double u = 0.0;
    if (part == 0)
        u = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    // ---- the hop: y_m = u_m - B_m^T y_jp ----
    double* ysj = ys[0];   // (every earlier use of ys[] is behind the barriers above)
    if (has_parent)
        receive(jp, ysj);
    else if (part == 0)
        ysj[c] = 0.0;
    __syncthreads();   // also: everyone has read red[] above
    CH_RT(m, 1);
    double a3 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        a3 += sB[r][tid] * ysj[part * 16 + r];
    red[part][c] = a3;
    __syncthreads();
    if (part == 0)
        publish(u - ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])));
    CH_RT(m, 2);
    if (tid == 0) {
        if (s_timeout)
            raise_sync_timeout(ctl, 2);
        chain_block_done(n_done, n_blk, epoch_word, epoch, nz != nullptr, m);
    }
}

// Explicit inverse of one 64x64 lower-triangular diagonal factor (for the chained back-substitution,
// which then needs a 64x64 GEMV per block instead of 16 dependent 4x4 solves).  One wave, thread c
// solves L x = e_c; entries above row c are zero, so all lanes run the same 2016 multiply-adds.
// Runs as one extra workgroup of a later launch, beside the latency-bound panel: free.
// L (row stride kLd, upper part zero) and di (reciprocal diagonal) already in LDS; wave 0 computes.
__device__ __forceinline__ void chol_inverse_lds(const double* L, const double* di, double* __restrict__ Linvk)
{
    const int tid = threadIdx.x;
    if (tid >= 64)
        return;
    const int c = tid;
    double x[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int m = 0; m + 3 < i; m += 4) {
            s0 += L[i * kLd + m] * x[m];
            s1 += L[i * kLd + m + 1] * x[m + 1];
            s2 += L[i * kLd + m + 2] * x[m + 2];
            s3 += L[i * kLd + m + 3] * x[m + 3];
        }
#pragma unroll
        for (int m = (i / 4) * 4; m < i; ++m)
            s0 += L[i * kLd + m] * x[m];
        const double s = (s0 + s1) + (s2 + s3);
        x[i] = (i == c) ? di[i] : ((i < c) ? 0.0 : -s * di[i]);
        Linvk[i * 64 + c] = x[i];
    }
}

__device__ __forceinline__ void chol_inverse_wg(const double* __restrict__ Ldk, const double* __restrict__ dinvk,
                                                double* __restrict__ Linvk, double* smem)
{
    double* L = smem;
    double* di = smem + 64 * kLd;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        L[r * kLd + c] = (c <= r) ? Ldk[idx] : 0.0;
    }
    if (tid < 64)
        di[tid] = dinvk[tid];
    __syncthreads();
    chol_inverse_lds(L, di, Linvk);
}

// ------------------------------------------------------------------------------------------------
// Dataflow factorisation (k_chol_dataflow): the whole Cholesky of a reduced system of up to 21 blocks (order
// <= 1344: 224 kept poses) in ONE launch.  Every (block column j, row block R > j) pair is a workgroup that keeps
// its two 64x64 blocks -- a replica of the diagonal block (j,j) and the block (R,j) -- in MFMA accumulators from
// the first to the last instruction (left-looking): it first subtracts the contributions of the panels k < j,
// eight columns at a time, as those columns are published by the workgroups (k,j) and (k,R), then factors its own
// panel (eight rounds of eight columns: 8x8 pivot block, rows scaled in the same round, rank-8 update) and
// publishes its scaled columns round by round.  One more workgroup per block column holds only the diagonal
// block; it writes the diagonal factor, its reciprocals and its inverse (back-substitution, covariance).
//
// A published slice (64 rows x 8 columns) travels as self-validating {epoch, 32 value bits} granules
// (cdna_hip_programming.md Guideline 16, R2: the data is the flag): layout [half][column][row], written by ONE
// wave with one aligned agent-scope 8-byte store per granule, swept by the consumers with agent-scope loads
// until every tag carries this factorisation's epoch.  Nothing else is shared inside the launch: S is read at
// the start (written by the previous kernel) and L, Ld, dinv, Linv are written for the kernels that follow.
//
// What this buys at 19 blocks against one k_chol_step launch per block column (21 us each): the accumulators
// never leave the registers between panels (no load / rank-64 update / store per step: 9.4 us), the update of
// column j+1 by panel j is applied eight columns behind the panel's own rounds on OTHER compute units, and the
// 19 launch boundaries go.  Measured time line (tools/gpu_df_stamps.sh): 1.05-1.2 us per 8-column round (the
// pivot chain: 8x8 Cholesky ~1200 cycles + row scaling ~750 + hand-offs), ~3 us from the last round of a block
// column to the first pivot block of the next (granule latency + the consumer's backlog: a slice costs a worker
// 26 MFMAs = 0.7 us + operand loads, about the rate at which slices are produced), 11.7 us per block column.
//
// Inside a workgroup the waves are specialised:
//   wave 0 (P0) factors the 8x8 pivot blocks and scales the rows of the diagonal block, nothing else;
//   wave 1 (P1) does the same for the rows below and publishes them (granules);
//   waves 2, 3 (W0, W1) own ALL accumulator tiles (13 each: W0 the 10 lower tiles of the diagonal block + 3 of
//     the block below, W1 the other 13) and issue every MFMA;
// each wave alone on its SIMD (f64 MFMA and f64 VALU share a SIMD's FP64 pipe: a wave that does both serialises
// them, two waves on two SIMDs do not).  A round is three barrier-separated phases:
//   1  W0: rank-8 update of the tile that holds the next pivot block, pivot block -> LDS
//   2  P0, P1: 8x8 Cholesky                          ||  W0, W1: update + publish the rest of the pivot tile
//                                                         column, then part of the remaining tiles
//   3  P0, P1: scale their rows, write them back      ||  W0, W1: the remaining tiles of the update
// so the update of round r hides behind the chain of round r+1.  While a workgroup consumes the panels before its
// own, P0/P1 sweep and stage the published slices and W0/W1 apply them.
//
// Variants built and measured at 19 blocks, then removed: every wave owning a 16-row strip
// of both blocks with waves 0/1 also carrying the pivot chain (233 us against 223 us); the pivot waves applying
// the previous round's rank-8 update to whole columns themselves so that a round is one barrier (rounds 1.45 us:
// 64 LDS reads + 64 FMAs per lane and round cost more than the wait they remove; 253 us); the pivot waves fixing
// up only the 8x8 pivot block from an early copy (230 us: the workers' slice backlog, not the pivot chain, sets
// the pace).
//
// Progress: blockIdx is panel-major, so a workgroup only waits for workgroups with smaller blockIdx; with the
// in-order dispatch observed on this hardware the earliest unfinished workgroup is always resident and never
// waits for an undispatched one, whatever the residency (the launcher still only uses this kernel when all
// workgroups fit on the chip at one per CU).  Every spin is bounded and a timeout raises an abort word that
// ends every other spin; the factorisation is then reported as failed (NaN poisoning + lin_fail).
// ------------------------------------------------------------------------------------------------
constexpr int kDfSlice = 2 * 8 * 64;         // granules (8 bytes each) per published slice
// LDS row stride (doubles) of the dataflow kernel's 64x8 panel buffers and 8x8 blocks: EVEN, so a row starts 16-byte
// aligned and is read / written two entries per instruction (the pivot waves' LDS round trips are on the critical path:
// 4 instead of 8 per row); 10: a 16-lane group of ds_read_b128 covers all 64 banks, the workers' ds_read_b64 operand
// fetches (16 rows x 2 columns per half wave) stay conflict-free
constexpr int kPsD = 10;
constexpr unsigned kDfSpinDefault = 1u << 21;   // polls of ~0.3 us each before giving up
constexpr int kDfXs = 8 * kLdsRow;            // doubles per staged slice, k-major [8][kLdsRow]
// doubles: 73 KB used, declared as 84 KB.  The copy of the diagonal factor that the block inverse reads (64 x kLd + 64,
// diagonal-only role, after the last round) lives in the panel / slice buffers, which are dead by then.  84 KB: two
// of these workgroups never share a CU (every wave alone on its SIMD), while a rank-k update workgroup (72 KB) still
// fits on the same CU beside a factorisation workgroup that is waiting for its block column (156 of 160 KB).
constexpr int kDfSmemUsed = 64 * kLdT + 4 * 64 * kPsD + 64 + 2 * 8 * kPsD + 4 * kDfXs;
constexpr int kDfSmem = 84 * 1024 / 8;
static_assert(kDfSmemUsed <= kDfSmem, "dataflow LDS layout");
static_assert(4 * 64 * kPsD + 64 + 4 * kDfXs >= 64 * kLd + 64, "the inverse's staging area must fit into the dead buffers");

struct DfArgs {
    LmCtl* ctl;
    double* S;
    int ld, n_pad, n_blk;
    double* dinv;
    double* Ld;
    double* Linv;
    unsigned long long* G;       // [n_blk (n_blk + 1) / 2][8][kDfSlice]
    const unsigned* epoch_word;  // bumped by the back-substitution chain that follows
    unsigned* abort_word;        // == epoch: some workgroup gave up waiting
    unsigned spin_limit;         // polls before a wait gives up (set per launch from LmCtl::spin_limit_df)
    const unsigned long long* nz;   // block structure of the factor: bit k of row i (kDfMaskWords words per row, up to 255
                                    // block columns) = L(i, k) may be non-zero (after fill); null: dense.  A workgroup then only
                                    // consumes the panels its row and column share, a structurally zero tile has no workgroup
                                    // and no slot for its slices (tree orderings)
    const unsigned char* order;     // with nz: [n_blk][kDfMaxBlk] the panels of block column j in the order they are expected
                                    // to be finished (a column of a separator takes the panels of the subtree that is done
                                    // first first, instead of waiting for panel 9 with panels 12-14 already there)
    const int32_t* wg;              // with nz: [gridDim.x][2] (block column, block row) of every workgroup, panel-major: the
                                    // non-zero blocks below the diagonal of a column (the right-hand side row last), then the
                                    // diagonal-only workgroup
    const int32_t* slot;            // with nz: [n_blk][n_blk + 1] slot of block (k, rb)'s slices in G, -1: structurally zero
    double* Gc;                     // BULK kernels: [slot][64 columns][64 rows] the block once more, as plain doubles, written when
    unsigned* done;                 // the workgroup is finished; done[slot] == epoch says so (release / acquire, agent scope)
};



// number of panels k < j that block column j of the factor has an entry in (dense: all of them)
__device__ __forceinline__ int df_num_panels(const DfArgs& a, const int j)
{
    if (!a.nz)
        return j;
    int n = 0;
#pragma unroll
    for (int w = 0; w < kDfMaskWords; ++w) {
        const int lo = 64 * w;
        if (j <= lo)
            break;
        const unsigned long long below = (j - lo >= 64) ? ~0ull : ((1ull << (j - lo)) - 1ull);
        n += __popcll(a.nz[kDfMaskWords * j + w] & below);
    }
    return n;
}

__device__ __forceinline__ double df_value(const unsigned long long lo, const unsigned long long hi)
{
    return __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
}

#ifdef VMM_STAMPS
__device__ unsigned long long g_df_stamps[32][128];   // [block column][slot]: s_memrealtime (100 MHz) / s_memtime
#define DF_RT(slot)                                                                              \
    do {                                                                                         \
        if (stamp_on && lane == 0)                                                               \
            g_df_stamps[stamp_j][slot] = __builtin_amdgcn_s_memrealtime();                       \
    } while (0)
#define DF_CY(slot)                                                                              \
    do {                                                                                         \
        if (stamp_cy && lane == 0)                                                               \
            g_df_stamps[stamp_j][(slot) - stamp_off] = __builtin_amdgcn_s_memtime();             \
    } while (0)
extern "C" int vmm_ba_debug_read_df_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_stamps), sizeof(unsigned long long) * 32 * 128);
}
#else
#define DF_RT(slot)
#define DF_CY(slot)
#endif

namespace df2 {

// chol8 (above) for the dataflow kernel: the block at D has row stride kPsD and is read two entries at a time; the
// validity test is off the chain altogether -- a non-positive or non-finite pivot gives NaN (v_rsq_f64 of a negative
// number, 0 * inf in the correction), every later entry of the factor inherits it, and ok is read off the last reciprocal
__device__ __forceinline__ void chol8_df(const double* __restrict__ D, Piv8& p)
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c <= r; c += 2) {
            const double2 v = *reinterpret_cast<const double2*>(D + r * kPsD + c);
            p.l[tri8(r, c)] = v.x;
            if (c + 1 <= r)
                p.l[tri8(r, c + 1)] = v.y;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double t = p.l[tri8(j, j)];
        const double y0 = __builtin_amdgcn_rsq(t);
        const double e = fma(-t * y0, y0, 1.0);
        const double inv = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        p.inv[j] = inv;
        p.l[tri8(j, j)] = t * inv;
#pragma unroll
        for (int i = j + 1; i < 8; ++i)
            p.l[tri8(i, j)] *= inv;
#pragma unroll
        for (int c = j + 1; c < 8; ++c)
#pragma unroll
            for (int i = c; i < 8; ++i)
                p.l[tri8(i, c)] = fma(-p.l[tri8(i, j)], p.l[tri8(c, j)], p.l[tri8(i, c)]);
    }
    p.ok = isfinite(p.inv[7]);
}

// tile tables: worker 0 = D lower tiles (row-major) + T(0,0..2); worker 1 = T(0,3) + T(1..3, 0..3)
__device__ __forceinline__ constexpr bool is_t(int wk, int i) { return wk == 0 ? i >= 10 : true; }
__device__ __forceinline__ constexpr int tile_i(int wk, int i)
{
    if (wk == 0)
        return i >= 10 ? 0 : (i >= 6 ? 3 : (i >= 3 ? 2 : (i >= 1 ? 1 : 0)));
    return i == 0 ? 0 : 1 + (i - 1) / 4;
}
__device__ __forceinline__ constexpr int tile_j(int wk, int i)
{
    if (wk == 0)
        return i >= 10 ? i - 10 : i - tile_i(0, i) * (tile_i(0, i) + 1) / 2;
    return i == 0 ? 3 : (i - 1) % 4;
}
// Round r factors columns J0 = 8 r .. J0 + 7 of the block column.  The pivot waves form the NEXT pivot block themselves
// (pivot_round: the 8x8 Gram product of the eight scaled rows below the pivot block), so what the pivot chain needs from
// the workers before it can scale its rows is: the columns of this round for all rows below the pivot block (the tiles of
// the pivot tile column tc = r >> 1) and the diagonal 8x8 block of round r + 1 as it is BEFORE this round's update (the
// Gram product is subtracted from it).  That block sits in the pivot tile for even r and in the next diagonal tile for
// odd r.
// Phase of tile i in the rank-8 update with the columns of round r - 1, applied during round r:
//   0  not touched (left of the pivot tile column; the diagonal tile of an odd round: what is left of it is the pivot
//      block the pivot waves compute themselves)
//   1  round 0 only: the tile that holds the first pivot block (one more barrier: the 8x8 Cholesky starts behind it)
//   2  needed by the pivot waves before they scale, then as many of the others as fit beside the 8x8 Cholesky
//   3  the others, beside the scaling
#ifndef VMM_DF_FILL2
#define VMM_DF_FILL2 4
#endif
constexpr int kFill2 = VMM_DF_FILL2;
__device__ __forceinline__ constexpr bool urgent_tile(int wk, int i, int r)
{
    const int tc = r >> 1, tj = tile_j(wk, i), ti = tile_i(wk, i);
    const bool diag = !is_t(wk, i) && ti == tj;
    if (tj == tc)
        return !(diag && (r & 1));
    return (r & 1) && diag && tj == tc + 1;
}
__device__ __forceinline__ constexpr int phase_of(int wk, int i, int r, bool has_t)
{
    if (is_t(wk, i) && !has_t)
        return 0;
    const int tc = r >> 1, tj = tile_j(wk, i), ti = tile_i(wk, i);
    const bool diag = !is_t(wk, i) && ti == tj;
    if (tj < tc || (tj == tc && diag && (r & 1)))
        return 0;
    if (urgent_tile(wk, i, r))
        return (r == 0 && diag) ? 1 : 2;
    // remaining tiles: fill phase 2 up to kFill2 tiles per worker: operand loads + 2 MFMAs per tile + the publication of
    // the urgent ones must end before the 8x8 Cholesky beside them does (~1100 cycles), or the pivot chain waits
    int n_urgent = 0, rank = 0;
    for (int k = 0; k < 13; ++k) {
        if (is_t(wk, k) && !has_t)
            continue;
        if (urgent_tile(wk, k, r))
            ++n_urgent;
        else if (tile_j(wk, k) >= tc && !(tile_j(wk, k) == tc) && k < i)
            ++rank;
    }
    return (n_urgent + rank < kFill2) ? 2 : 3;
}

struct Ops {   // MFMA operands of one k-step: A of the diagonal block's tile rows, A of the block below, B
    double ad[4], at[4], b[4];
};

// operands of the rank-8 update with the scaled columns in pd / pt (row-major, stride kPsD); rows < m are masked
template <int WK, bool HAS_T>
__device__ __forceinline__ void load_ops_panel(const double* pd, const double* pt, const int m, const int fr, const int fk,
                                               Ops (&o)[2])
{
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * t + fr;
            const double v = pd[row * kPsD + 4 * ks + fk];
            const double vm = (row >= m) ? v : 0.0;
            o[ks].b[t] = vm;
            o[ks].ad[t] = (WK == 0) ? -vm : 0.0;
            o[ks].at[t] = (HAS_T && (WK == 1 || t == 0)) ? -pt[row * kPsD + 4 * ks + fk] : 0.0;
        }
}

// operands from staged slices (k-major, stride kLdsRow)
template <int WK, bool HAS_T>
__device__ __forceinline__ void load_ops_slice(const double* XJ, const double* XR, const int fr, const int fk, Ops (&o)[2])
{
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = (4 * ks + fk) * kLdsRow + 16 * t + fr;
            const double v = XJ[off];
            o[ks].b[t] = v;
            o[ks].ad[t] = (WK == 0) ? -v : 0.0;
            o[ks].at[t] = (HAS_T && (WK == 1 || t == 0)) ? -XR[off] : 0.0;
        }
}

template <int WK, int I>
__device__ __forceinline__ void mfma_tile(double4_t (&acc)[13], const Ops (&o)[2])
{
    constexpr int ti = tile_i(WK, I), tj = tile_j(WK, I);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const double a = is_t(WK, I) ? o[ks].at[ti] : o[ks].ad[ti];
        acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, o[ks].b[tj], acc[I], 0, 0, 0);
    }
}

// columns cj..cj+7 of tile I -> the panel buffers (rows of the diagonal block -> pd, rows below -> pt)
template <int WK, int I>
__device__ __forceinline__ void publish_tile(const double4_t (&acc)[13], double* pd, double* pt, const int cj, const int fr,
                                             const int fk)
{
    constexpr int ti = tile_i(WK, I);
    if (fr >= cj && fr < cj + kPw) {
        double* dst = (is_t(WK, I) ? pt : pd) + (16 * ti + fk) * kPsD + (fr - cj);
        dst[0] = acc[I][0];
        dst[4 * kPsD] = acc[I][1];
        dst[8 * kPsD] = acc[I][2];
        dst[12 * kPsD] = acc[I][3];
    }
}

// 8x8 quadrant (QR, QC) of tile I -> dst (row stride kPsD): the first pivot block of a panel and the diagonal block the
// pivot waves subtract their Gram product from
template <int I, int QR, int QC>
__device__ __forceinline__ void publish_quadrant(const double4_t (&acc)[13], double* dst, const int fr, const int fk)
{
    if (fr >= 8 * QC && fr < 8 * QC + 8) {
        double* d = dst + fk * kPsD + (fr - 8 * QC);
        d[0] = acc[I][2 * QR];
        d[4 * kPsD] = acc[I][2 * QR + 1];
    }
}

// one phase of a worker in round R8 (columns 8 R8 ..): the tiles of that phase are updated (UPDATE: not in the first round
// of a panel, whose accumulators are complete); the tiles of the pivot tile column are published, and so are the first
// pivot block (round 0, -> pb) and the diagonal block of the next round before this round's update (-> nd)
template <int WK, bool HAS_T, int R8, int PHASE, bool UPDATE, int... Is>
__device__ __forceinline__ void worker_phase(double4_t (&acc)[13], const Ops (&o)[2], double* pd, double* pt, double* pb,
                                             double* nd, const int fr, const int fk, std::integer_sequence<int, Is...>)
{
    constexpr int tc = R8 >> 1;
    // all MFMAs of the phase first, the urgent tiles leading: a publication right behind its own tile's MFMAs would wait
    // for the matrix pipeline to drain once per tile
    auto upd = [&](auto idx, auto urgent_pass) {
        constexpr int I = decltype(idx)::value;
        constexpr bool U = decltype(urgent_pass)::value;
        if constexpr (UPDATE && phase_of(WK, I, R8, HAS_T) == PHASE && urgent_tile(WK, I, R8) == U)
            mfma_tile<WK, I>(acc, o);
    };
    (upd(std::integral_constant<int, Is>{}, std::true_type{}), ...);
    (upd(std::integral_constant<int, Is>{}, std::false_type{}), ...);
    auto pub = [&](auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (phase_of(WK, I, R8, HAS_T) == PHASE) {
            constexpr int ti = tile_i(WK, I), tj = tile_j(WK, I);
            constexpr bool diag = !is_t(WK, I) && ti == tj;
            if constexpr (tj == tc)
                publish_tile<WK, I>(acc, pd, pt, (8 * R8) & 15, fr, fk);
            if constexpr (diag && R8 == 0 && tj == 0)
                publish_quadrant<I, 0, 0>(acc, pb, fr, fk);
            if constexpr (diag && R8 < 7 && !(R8 & 1) && tj == tc)
                publish_quadrant<I, 1, 1>(acc, nd, fr, fk);
            if constexpr (diag && R8 < 7 && (R8 & 1) && tj == tc + 1)
                publish_quadrant<I, 0, 0>(acc, nd, fr, fk);
        }
    };
    (pub(std::integral_constant<int, Is>{}), ...);
}

template <int WK, bool HAS_T, int... Is>
__device__ __forceinline__ void worker_apply_slice(double4_t (&acc)[13], const Ops (&o)[2], std::integer_sequence<int, Is...>)
{
    auto one = [&](auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (HAS_T || !is_t(WK, I))
            mfma_tile<WK, I>(acc, o);
    };
    (one(std::integral_constant<int, Is>{}), ...);
}

using Seq13 = std::make_integer_sequence<int, 13>;

template <typename F, int... Is>
__device__ __forceinline__ void for_tiles(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

// a pivot wave's whole slice: sixteen granules per lane (lane = row)
struct SliceRegs {
    unsigned long long lo[8], hi[8];
};

__device__ __forceinline__ void issue_slice(const unsigned long long* sl, const int lane, SliceRegs& g)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        g.lo[q] = __hip_atomic_load(sl + q * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g.hi[q] = __hip_atomic_load(sl + 512 + q * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ bool slice_valid(const SliceRegs& g, const unsigned epoch)
{
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        ok = ok && (unsigned)(g.lo[q] >> 32) == epoch && (unsigned)(g.hi[q] >> 32) == epoch;
    return ok;
}

// direct: the slice is expected any moment (the panel right before mine): sweep it again instead of probing one
// granule first
__device__ __forceinline__ bool wait_slice(const unsigned long long* sl, const int lane, const unsigned epoch,
                                           const unsigned* abort_word, const bool direct, SliceRegs& g,
                                           const unsigned kDfSpinLimit, bool* spun = nullptr)
{
    for (unsigned n = 0;;) {
        if (kDfSpinLimit != 1u && __all(slice_valid(g, epoch)))   // a limit of 1 (debugging) gives up even on valid data
            return true;
        if (spun)
            *spun = true;   // the first look came back stale: this slice was not there yet
        if (!direct) {
            for (;;) {
                const unsigned long long pv
                    = __hip_atomic_load(sl + 512 + 7 * 64 + 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(pv >> 32) == epoch)
                    break;
                if (++n >= kDfSpinLimit)
                    return false;
                if ((n & 63u) == 0u
                    && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)
                    return false;
                __builtin_amdgcn_s_sleep(2);
            }
        } else {
            if ((n & 63u) == 63u
                && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)
                return false;
            __builtin_amdgcn_s_sleep(1);
        }
        if (++n >= kDfSpinLimit)
            return false;
        issue_slice(sl, lane, g);
    }
}

// The slices a workgroup waits for while their producer is still at work (the panel right before mine).  A look at a slice
// is a round trip to the level the XCDs share (~1.1 us under this kernel's traffic) and the producer publishes one every
// ~0.85 us, so ONE look at a time cannot keep up: the look at slice s+1 must be on its way before slice s has been seen.
// Both the slice waited for (g) and the next one (gn, requested ahead by the caller) are looked at again each time their
// previous look comes back stale, alternately, so each is sampled once per round trip, half a round trip apart, and the
// next slice is usually complete in its registers when the current one has been staged.
// (Measured before: the copy requested two slices ahead was always stale, every slice then cost a fresh round trip after its
// predecessor, and each block column started 2.3 us behind the last slice of the previous one, 3 us with shorter rounds.)
__device__ __forceinline__ bool wait_slice_pair(const unsigned long long* sl, const unsigned long long* sl_next, const int lane,
                                                const unsigned epoch, const unsigned* abort_word, SliceRegs& g, SliceRegs& gn,
                                                const unsigned kDfSpinLimit, bool* spun = nullptr)
{
    for (unsigned n = 0;;) {
        if (kDfSpinLimit != 1u && __all(slice_valid(g, epoch)))   // a limit of 1 (debugging) gives up even on valid data
            return true;
        if (spun)
            *spun = true;
        if (++n >= kDfSpinLimit)
            return false;
        if ((n & 63u) == 63u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)
            return false;
        issue_slice(sl, lane, g);
        if (sl_next && !__all(slice_valid(gn, epoch)))
            issue_slice(sl_next, lane, gn);
    }
}

struct Lds {
    double* RA;     // D-only role: L^T (stride kLdT); others: result tile (stride kLd)
    double* Pd;     // [2][64][kPsD] panel columns of the diagonal block's rows (ping-pong between rounds)
    double* Pt;     // the same for the rows below
    double* invd;
    double* Pb;     // [8][kPsD] the pivot block of the current round (round 0: from the workers; then from the pivot waves)
    double* Nd;     // [8][kPsD] the diagonal block of the next round before this round's update (from the workers)
    double* Xs;     // [2 buffers][J | R][8][kLdsRow] staged slices of earlier panels
    int stamp_j;    // diagnostic build: block column whose (j, j+1) workgroup records time stamps (else -1)
};

struct SliceMap {
    unsigned long long* G;
    int n_blk;
    const int32_t* slot;   // tree orderings: [n_blk][n_blk + 1] slot of block (k, rb), only the non-zero blocks have one
    __device__ __forceinline__ int64_t index(int k, int rb) const
    {
        return slot ? (int64_t)slot[k * (n_blk + 1) + rb] : (int64_t)k * n_blk - (int64_t)k * (k - 1) / 2 + (rb - k - 1);
    }
    __device__ __forceinline__ unsigned long long* at(int k, int rb, int r) const
    {
        return G + (index(k, rb) * 8 + r) * kDfSlice;
    }
};

// ---- the pivot waves' program: sweeps during the earlier panels, then 8 x (8x8 Cholesky, scale rows, next pivot block) ----
// Barriers: one per consumed slice, then per round (A, round 0 only) B, C -- the same sequence as worker_path.
//   A  the first pivot block of the panel is in Pb (from the workers' accumulators)
//   B  all rows below the pivot block, columns J0..J0+7, are in pdc / ptc and the next diagonal block in Nd
//   C  the scaled columns are in pdc / ptc, the next pivot block in Pb
// The pivot chain is 8x8 Cholesky -> B -> scale the rows -> next pivot block = Nd - X X^T for the eight scaled rows X right
// below the pivot block (wave 0, lane = one entry of the block, the rows exchanged through pdc: same wave, no barrier)
// -> C -> 8x8 Cholesky; the workers' rank-8 update of the pivot tile column runs beside the 8x8 Cholesky instead of in
// front of it (until round 3 this was a third phase of ~640 cycles per round: MFMA update of the pivot tile, LDS, barrier).
template <int J0, bool HAS_T>
__device__ __forceinline__ void pivot_round(const int w, const int lane, const Lds& m, bool& ok, unsigned long long* gs,
                                            const unsigned epoch)
{
    double* pdc = m.Pd + ((J0 >> 3) & 1) * 64 * kPsD;
    double* ptc = m.Pt + ((J0 >> 3) & 1) * 64 * kPsD;
    const bool active = w == 0 || HAS_T;
#ifdef VMM_STAMPS
    const bool stamp_on = m.stamp_j >= 0 && w == 0;
    const bool stamp_cy = m.stamp_j >= 0;
    const int stamp_off = w == 0 ? 0 : 8;
    const int stamp_j = m.stamp_j;
#endif
    if (J0 == 0)
        __syncthreads();   // A: the first pivot block is in Pb
    if (J0 == 16) DF_CY(40);
    Piv8 p;
    if (active) {
        chol8_df(m.Pb, p);
        // the factor is complete BEFORE the barrier: left alone, the compiler sinks its arithmetic behind the barrier
        // and the 8x8 Cholesky no longer overlaps with the workers' phase 2 (measured with the stamps build)
#pragma unroll
        for (int k = 0; k < 36; ++k)
            asm volatile("" : "+v"(p.l[k]));
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("" : "+v"(p.inv[k]));
    }
    if (J0 == 16) DF_CY(41);
    __syncthreads();   // B: all rows of columns J0..J0+7 are in pdc / ptc, the next diagonal block in Nd
    if (J0 == 16) DF_CY(42);
    if (active) {
        double* row = (w == 0 ? pdc : ptc) + lane * kPsD;
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const double2 v = *reinterpret_cast<const double2*>(row + q);
            x[q] = v.x;
            x[q + 1] = v.y;
        }
        if (w == 1) {
            scale8(x, p);   // x = a L8^{-T}
            // the rows below leave for the other workgroups first (the longest latency of the round; holding them
            // back behind the barrier in all rounds but the last was measured slower) ...
            const unsigned long long tag = (unsigned long long)epoch << 32;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(x[q]);
                __hip_atomic_store(gs + q * 64 + lane, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gs + 512 + q * 64 + lane, tag | (bits >> 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < 8; q += 2)
                *reinterpret_cast<double2*>(row + q) = make_double2(x[q], x[q + 1]);
        } else {
            ok = ok && p.ok;
            // Wave 0 is the pivot chain.  The next pivot block: entry (gi, gj) = Nd - sum_q X[gi][q] X[gj][q] over the
            // scaled rows X = rows J0+8 .. J0+15, which lanes J0+8 .. J0+15 of this very wave produce (LDS operations of a
            // wave stay in order: no barrier).  Column q of a row is final after step q of the scaling, so it is written
            // and the two entries of it a lane needs are requested back right there: the LDS round trips run beside the
            // remaining steps instead of behind the last one.  Every lane writes its row -- rows up to the pivot block hold
            // nothing anybody reads (the workers mask them, load_ops_panel).
            constexpr bool NEXT = J0 + kPw < 64;
            const int gi = lane >> 3, gj = lane & 7;
            const double* xi = pdc + (J0 + kPw + gi) * kPsD;
            const double* xj = pdc + (J0 + kPw + gj) * kPsD;
            double sacc = NEXT ? m.Nd[gi * kPsD + gj] : 0.0;
            double vi[8], vj[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                x[q] *= p.inv[q];
#pragma unroll
                for (int c = q + 1; c < 8; ++c)
                    x[c] = fma(-x[q], p.l[tri8(c, q)], x[c]);
                row[q] = x[q];
                if (NEXT) {
                    vi[q] = xi[q];
                    vj[q] = xj[q];
                }
            }
#ifdef VMM_STAMPS
#pragma unroll
            for (int q = 0; q < 8; ++q)
                asm volatile("" : "+v"(x[q]));
            if (J0 == 16) DF_CY(56);
#endif
            if (NEXT) {
                if (J0 == 16) DF_CY(57);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    sacc = fma(-vi[q], vj[q], sacc);
#ifdef VMM_STAMPS
                asm volatile("" : "+v"(sacc));
                if (J0 == 16) DF_CY(58);
#endif
                m.Pb[gi * kPsD + gj] = sacc;
            }
        }
        if (J0 == 16) DF_CY(43);
        __syncthreads();   // C: the scaled columns are in pdc / ptc, the next pivot block in Pb
        if (J0 == 16) DF_CY(44);
        DF_RT(2 + (J0 >> 3));
        // ... what only this workgroup's final write-back needs is stored behind the barrier, beside the next 8x8 Cholesky
        if (w == 1) {
            double* rr = m.RA + lane * kLd + J0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                rr[q] = x[q];
        } else if (!HAS_T) {
            // keep L^T for the write-back: x below the pivot block, the factor inside, zero above
            const int r = lane - J0;
            const bool below = r >= kPw, above = r < 0;
            double* At = m.RA;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double inside = 0.0;
#pragma unroll
                for (int rr = q; rr < 8; ++rr)
                    inside = (r == rr) ? p.l[tri8(rr, q)] : inside;
                At[(J0 + q) * kLdT + lane] = below ? x[q] : (above ? 0.0 : inside);
            }
            if (r >= 0 && r < kPw) {
                double iv = 0.0;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
                    iv = (r == rr) ? p.inv[rr] : iv;
                m.invd[lane] = iv;
            }
        }
    } else {
        __syncthreads();   // C (idle pivot wave of the diagonal-only role)
    }
}

// TREE: the factor has a block structure (DfArgs::nz, tree orderings): only the panels this block column depends on are
// consumed, in DfArgs::order.  !TREE is the dense kernel: panels 0 .. j-1 in ascending order.
// BULK: panels that are COMPLETE when this workgroup gets to them -- it works off a backlog: a separator column of a tree
// ordering, a late block column of a system with more workgroups than compute units -- are read from the producers' compact
// copies (DfArgs::Gc: plain doubles behind a completion word) instead of swept as granules: half the bytes and half the
// loads of a look, no validity test, two slices per register set, requested across panel boundaries.  A panel still in
// production is tracked through its granules as before; once a look has come back stale the workgroup has caught up
// with production and stops asking for completion words.  !BULK is the kernel of round 4, instruction for instruction.
// MODE bit 1 (HELP): the workgroup has six waves -- two more workers (waves 4, 5) on the pivot waves' SIMDs, which hold six of
// each worker's thirteen tiles while the EARLIER panels are applied (the pivot waves only sweep then: loads and integer
// work, nothing on the f64 pipe an MFMA of another wave would block) and hand them to the workers through the LDS right
// before the last slice, where they end.  A tile sees the same MFMAs in the same order whoever issues them: the same bits.
template <bool HAS_T, bool TREE, int MODE>
__device__ __forceinline__ void pivot_path(const DfArgs& a, const int w, const int lane, const int j, const int R,
                                           const Lds& m, const SliceMap& sm, const unsigned epoch, int* s_timeout, bool& ok)
{
    constexpr bool BULK = (MODE & 1) != 0, HELP = (MODE & 2) != 0;
    const int n_it = TREE ? 8 * df_num_panels(a, j) : 8 * j;   // a multiple of 8
    if (n_it > 0) {
        // Two slices are on their way at any time (two register sets): a slice read costs a round trip to the level all
        // XCDs share (~1.0-1.3 us) and with one request in flight that was the pace of a workgroup working off panels that are
        // long complete -- slower than they are produced since the rounds got shorter, so every block column started later
        // behind its predecessor than the one before
        SliceRegs ga, gb;
        const bool sweeper = w == 0 || HAS_T;
        const int my_rb = (w == 0) ? j : R;
        const unsigned char* const ord = TREE ? a.order + kDfMaxBlk * j : nullptr;
        // TREE: wave 1 sweeps the slices of block row R; where L(R, k) is structurally zero nobody publishes one -- zeros
        const bool mine_all = !TREE || w == 0 || R >= a.n_blk;
        // What a tree ordering keeps in tables in global memory -- which panel comes at position it >> 3 of this block column's
        // list, whether block row R has an entry in it, where block (k, my_rb) publishes its slices -- is looked up once per
        // PANEL (two panels are in use around a panel boundary), not once per slice: three dependent loads in front of every
        // request cost ~15 % of the slice rate.
        struct PanelInfo {
            int pos, k;
            bool has;
            unsigned long long* base;
        };
        PanelInfo c0{ -1, 0, false, nullptr }, c1{ -1, 0, false, nullptr };
        auto panel_at = [&](const int it) -> const PanelInfo& {
            const int pos = it >> 3;
            if (!TREE) {   // dense: panel `pos`, every block there, its place is arithmetic
                c0.pos = c0.k = pos;
                c0.has = true;
                c0.base = sm.at(pos, my_rb, 0);
                return c0;
            }
            if (pos == c0.pos)
                return c0;
            if (pos == c1.pos)
                return c1;
            c1 = c0;
            c0.pos = pos;
            c0.k = TREE ? (int)ord[pos] : pos;
            c0.has = mine_all || nz_bit(a.nz, R, c0.k);
            c0.base = c0.has ? sm.at(c0.k, my_rb, 0) : nullptr;
            return c0;
        };
        auto request = [&](const int it, SliceRegs& g) {
            if (sweeper && it < n_it) {
                const PanelInfo& pi = panel_at(it);
                if (pi.has)
                    issue_slice(pi.base + (it & 7) * kDfSlice, lane, g);
            }
        };
        // HELP: the helpers' tiles reach the workers behind one more barrier, right before the last slice
        auto help_before = [&](const int it) {
            if (HELP && it == n_it - 1)
                __syncthreads();
        };
        auto consume = [&](const int it, SliceRegs& g, SliceRegs& gn) {
            help_before(it);
            const PanelInfo pi = sweeper ? panel_at(it) : PanelInfo{ it >> 3, 0, true, nullptr };
            const int k = pi.k;
            const bool have = pi.has;
            if (sweeper) {
                // the panel expected last (dense: the one right before mine) is swept directly instead of probed
                const bool last_panel = TREE ? it + 8 >= n_it : k == j - 1;
                bool got = true;
                if (have && last_panel) {
                    // the next slice belongs to the same panel unless this is the panel's last one
                    const bool next_too = (it & 7) != 7;
                    unsigned long long* const sl = pi.base + (it & 7) * kDfSlice;
                    got = wait_slice_pair(sl, next_too ? sl + kDfSlice : nullptr, lane, epoch,
                                          a.abort_word, g, gn, a.spin_limit, nullptr);
                } else if (have) {
                    got = wait_slice(pi.base + (it & 7) * kDfSlice, lane, epoch, a.abort_word, false, g, a.spin_limit,
                                     nullptr);
                }
#ifdef VMM_STAMPS
                if (m.stamp_j >= 0 && w == 0 && lane == 0 && it >= n_it - 2)
                    g_df_stamps[m.stamp_j][12 + (it - (n_it - 2))] = __builtin_amdgcn_s_memrealtime();
                if (m.stamp_j >= 0 && w == 0 && lane == 0 && it >= n_it - 8)   // the last panel's slices, one by one
                    g_df_stamps[m.stamp_j][14 + (it - (n_it - 8))] = __builtin_amdgcn_s_memrealtime();
                if (m.stamp_j >= 0 && w == 1 && lane == 0 && it >= n_it - 8)
                    g_df_stamps[m.stamp_j][64 + (it - (n_it - 8))] = __builtin_amdgcn_s_memrealtime();
#endif
                double* X = m.Xs + (it & 1) * 2 * kDfXs + (w == 0 ? 0 : kDfXs);
                const double nan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double xv = got ? df_value(g.lo[q], g.hi[q]) : nan;
                    X[q * kLdsRow + lane] = (TREE && !have) ? 0.0 : xv;
                }
                if (!got && lane == 0) {
                    *s_timeout = 1;
                    __hip_atomic_store(a.abort_word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
#ifdef VMM_STAMPS
            if (m.stamp_j >= 0 && w == 0 && lane == 0 && it >= n_it - 8)
                g_df_stamps[m.stamp_j][80 + (it - (n_it - 8))] = __builtin_amdgcn_s_memrealtime();
#endif
        };
        if constexpr (!BULK) {
            request(0, ga);
            request(1, gb);
            for (int it = 0; it < n_it; it += 2) {
                consume(it, ga, gb);
                request(it + 2, ga);   // requested while the workers apply slice it
                consume(it + 1, gb, ga);
                request(it + 3, gb);
            }
        } else {
            const int n_pan = n_it >> 3;
            // one register set = two slices of a compact copy: lo[q] = column q of slice 2p, hi[q] = of slice 2p + 1
            auto issue_pair = [&](const double* cb, const int p, SliceRegs& g) {
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(cb) + p * 1024 + lane;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    g.lo[q] = __hip_atomic_load(src + q * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    g.hi[q] = __hip_atomic_load(src + 512 + q * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            auto stage = [&](const int it, const unsigned long long (&v)[8]) {
                help_before(it);
                double* X = m.Xs + (it & 1) * 2 * kDfXs + (w == 0 ? 0 : kDfXs);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    X[q * kLdsRow + lane] = __longlong_as_double((long long)v[q]);
                __syncthreads();
            };
            // Is block (k, my_rb)'s compact copy written?  The completion word of a panel is requested one panel ahead (at the
            // start of the panel in front of it), so that looking at it never waits: a panel that completes later than that is
            // taken through its granules like one that is still in production.
            auto flag_of = [&](const int pos, const double*& cb) -> unsigned {
                cb = nullptr;
                if (!sweeper || pos >= n_pan)
                    return epoch + 1u;
                const PanelInfo pi = panel_at(8 * pos);
                if (!pi.has)
                    return epoch + 1u;
                const int64_t si = sm.index(pi.k, my_rb);
                cb = a.Gc + si * 4096;
                return __hip_atomic_load(a.done + si, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            bool pre = false;               // the coming panel's first two pairs are already in ga / gb
            const double* cb_cur = nullptr;
            const double* cb_next = nullptr;
            unsigned fl_next = flag_of(0, cb_next);
            for (int pos = 0; pos < n_pan; ++pos) {
                const int it0 = 8 * pos;
                const unsigned fl = fl_next;
                cb_cur = cb_next;
                fl_next = flag_of(pos + 1, cb_next);   // on its way while this panel is applied
                const bool bulk = pre || fl == epoch;
                if (bulk) {
                    if (!pre) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        issue_pair(cb_cur, 0, ga);
                        issue_pair(cb_cur, 1, gb);
                    }
                    pre = false;
                    stage(it0 + 0, ga.lo);
                    stage(it0 + 1, ga.hi);
                    issue_pair(cb_cur, 2, ga);
                    stage(it0 + 2, gb.lo);
                    stage(it0 + 3, gb.hi);
                    issue_pair(cb_cur, 3, gb);
                    stage(it0 + 4, ga.lo);
                    stage(it0 + 5, ga.hi);
                    const bool nbulk = fl_next == epoch;   // (requested eight slices ago)
                    if (nbulk) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        issue_pair(cb_next, 0, ga);
                    }
                    stage(it0 + 6, gb.lo);
                    stage(it0 + 7, gb.hi);
                    if (nbulk) {
                        issue_pair(cb_next, 1, gb);
                        pre = true;
                    }
                } else {
                    // granules: the panel is in production, structurally zero for my block row (zeros are staged), or this
                    // wave only keeps the barriers
                    request(it0, ga);
                    request(it0 + 1, gb);
#pragma unroll 1
                    for (int it = it0; it < it0 + 8; it += 2) {
                        consume(it, ga, gb);
                        if (it + 2 < it0 + 8)
                            request(it + 2, ga);
                        consume(it + 1, gb, ga);
                        if (it + 3 < it0 + 8)
                            request(it + 3, gb);
                    }
                }
            }
        }
    }
#ifdef VMM_STAMPS
    const bool stamp_on = m.stamp_j >= 0 && w == 0;
    const int stamp_j = m.stamp_j;
#endif
    DF_RT(1);
    unsigned long long* g0 = HAS_T ? sm.at(j, R, 0) : sm.G;
    pivot_round<0, HAS_T>(w, lane, m, ok, g0, epoch);
    pivot_round<8, HAS_T>(w, lane, m, ok, g0 + 1 * kDfSlice, epoch);
    pivot_round<16, HAS_T>(w, lane, m, ok, g0 + 2 * kDfSlice, epoch);
    pivot_round<24, HAS_T>(w, lane, m, ok, g0 + 3 * kDfSlice, epoch);
    pivot_round<32, HAS_T>(w, lane, m, ok, g0 + 4 * kDfSlice, epoch);
    pivot_round<40, HAS_T>(w, lane, m, ok, g0 + 5 * kDfSlice, epoch);
    pivot_round<48, HAS_T>(w, lane, m, ok, g0 + 6 * kDfSlice, epoch);
    pivot_round<56, HAS_T>(w, lane, m, ok, g0 + 7 * kDfSlice, epoch);
}

// ---- a worker wave's program ----
// SLICE (round 0 of a block column > 0 only): the "previous round" is the last slice of the previous panel, staged
// at XJ / XR and not applied yet -- its update of the first pivot tile column comes first like any round's, so the
// pivot waves start on the panel 2 MFMAs after the slice has arrived instead of 26 + a round.
template <int WK, int J0, bool HAS_T, bool SLICE = false>
__device__ __forceinline__ void worker_round(const int lane, double4_t (&acc)[13], const Lds& m, const double* XJ = nullptr,
                                             const double* XR = nullptr)
{
    static_assert(!SLICE || J0 == 0, "only the first round takes a slice");
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int R8 = J0 >> 3;
    double* pdc = m.Pd + (R8 & 1) * 64 * kPsD;
    double* ptc = m.Pt + (R8 & 1) * 64 * kPsD;
    const double* pdp = m.Pd + ((R8 & 1) ^ 1) * 64 * kPsD;
    const double* ptp = m.Pt + ((R8 & 1) ^ 1) * 64 * kPsD;
    constexpr bool UPD = J0 > 0 || SLICE;
#ifdef VMM_STAMPS
    const bool stamp_cy = m.stamp_j >= 0;
    const int stamp_off = WK == 0 ? 0 : 24;
    const int stamp_j = m.stamp_j;
#endif
    Ops o[2];
    if (J0 == 16) DF_CY(48);
    if (SLICE)
        load_ops_slice<WK, HAS_T>(XJ, XR, fr, fk, o);
    else if (UPD)
        load_ops_panel<WK, HAS_T>(pdp, ptp, J0, fr, fk, o);
    if (J0 == 0) {
        worker_phase<WK, HAS_T, R8, 1, UPD>(acc, o, pdc, ptc, m.Pb, m.Nd, fr, fk, Seq13{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // A
        __builtin_amdgcn_sched_barrier(0);
    }
    worker_phase<WK, HAS_T, R8, 2, UPD>(acc, o, pdc, ptc, m.Pb, m.Nd, fr, fk, Seq13{});
    if (J0 == 16) DF_CY(49);
    // MFMAs touch no memory, so the compiler is free to sink them behind a barrier -- and did: the rest of a round's
    // update ran in front of the next round's urgent tiles, on the in-order matrix pipeline, ~700 cycles of every round
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // B
    __builtin_amdgcn_sched_barrier(0);
    if (J0 == 16) DF_CY(50);
    worker_phase<WK, HAS_T, R8, 3, UPD>(acc, o, pdc, ptc, m.Pb, m.Nd, fr, fk, Seq13{});
    if (J0 == 16) DF_CY(51);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // C
    __builtin_amdgcn_sched_barrier(0);
    if (J0 == 16) DF_CY(52);
}

constexpr int kHelpSplit = 7;   // HELP: a worker keeps its tiles 0..6 while earlier panels are applied, its helper holds 7..12

// one tile of a worker, straight from global memory in accumulator layout
template <int WK, int I, bool HAS_T>
__device__ __forceinline__ void load_tile(double4_t (&acc)[13], const double* __restrict__ S, const int ld, const int n_pad,
                                          const int K0, const int R0, const int fr, const int fk)
{
    constexpr int ti = tile_i(WK, I), tj = tile_j(WK, I);
    acc[I] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
    if constexpr (!is_t(WK, I)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[I][r] = S[(int64_t)(K0 + 16 * ti + fk + 4 * r) * ld + K0 + 16 * tj + fr];
    } else if constexpr (HAS_T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = R0 + 16 * ti + fk + 4 * r;
            const int grow = (row <= n_pad) ? row : n_pad;   // clamp: always in bounds
            const double tv = S[(int64_t)grow * ld + K0 + 16 * tj + fr];
            acc[I][r] = (row <= n_pad) ? tv : 0.0;
        }
    }
}

template <int WK, bool HAS_T, bool TREE, bool HELP = false>
__device__ __forceinline__ void worker_path(const DfArgs& a, const int lane, const int j, const int R, const Lds& m)
{
    const int fr = lane & 15, fk = lane >> 4;
    const int K0 = j * kNB, R0 = R * kNB;
    const int ld = a.ld, n_pad = a.n_pad;
    const double* __restrict__ S = a.S;
    const int n_it = TREE ? 8 * df_num_panels(a, j) : 8 * j;
    // accumulator tiles straight from global memory, in accumulator layout (HELP: the helper's tiles arrive later, unless
    // there is no earlier panel and hence no helper at work)
    double4_t acc[13];
    for_tiles([&](auto idx) {
        constexpr int I = decltype(idx)::value;
        if (!HELP || I < kHelpSplit || n_it == 0)
            load_tile<WK, I, HAS_T>(acc, S, ld, n_pad, K0, R0, fr, fk);
        else
            acc[I] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
    }, Seq13{});
    for (int it = 0; it + 1 < n_it; ++it) {
        const double* XJ = m.Xs + (it & 1) * 2 * kDfXs;
        const double* XR = XJ + kDfXs;
        __syncthreads();
        if (WK == 0 || HAS_T) {
            Ops o[2];
            load_ops_slice<WK, HAS_T>(XJ, XR, fr, fk, o);
            for_tiles([&](auto idx) {
                constexpr int I = decltype(idx)::value;
                if constexpr ((HAS_T || !is_t(WK, I)) && (!HELP || I < kHelpSplit))
                    mfma_tile<WK, I>(acc, o);
            }, Seq13{});
        }
#ifdef VMM_STAMPS
        if (m.stamp_j >= 0 && lane == 0 && it >= n_it - 8)
            g_df_stamps[m.stamp_j][(WK == 0 ? 72 : 88) + (it - (n_it - 8))] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    if (n_it > 0) {
        if (HELP) {
            __syncthreads();   // the helper's tiles are in the LDS (the result tile's area, dead until the rounds)
            const double* M = m.RA + WK * (13 - kHelpSplit) * 256;
            for_tiles([&](auto idx) {
                constexpr int I = decltype(idx)::value;
                if constexpr (I >= kHelpSplit) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[I][r] = M[((I - kHelpSplit) * 4 + r) * 64 + lane];
                }
            }, Seq13{});
        }
        const double* XJ = m.Xs + ((n_it - 1) & 1) * 2 * kDfXs;
        __syncthreads();   // the last slice is staged
        worker_round<WK, 0, HAS_T, true>(lane, acc, m, XJ, XJ + kDfXs);
    } else {
        worker_round<WK, 0, HAS_T>(lane, acc, m);
    }
    worker_round<WK, 8, HAS_T>(lane, acc, m);
    worker_round<WK, 16, HAS_T>(lane, acc, m);
    worker_round<WK, 24, HAS_T>(lane, acc, m);
    worker_round<WK, 32, HAS_T>(lane, acc, m);
    worker_round<WK, 40, HAS_T>(lane, acc, m);
    worker_round<WK, 48, HAS_T>(lane, acc, m);
    worker_round<WK, 56, HAS_T>(lane, acc, m);
}

// HELP: waves 4 and 5.  Worker WK's tiles kHelpSplit..12 from the start of the workgroup until the earlier panels are
// applied (all slices but the last one), then into the LDS for the worker, and out.
template <int WK, bool HAS_T, bool TREE>
__device__ __forceinline__ void helper_path(const DfArgs& a, const int lane, const int j, const int R, const Lds& m)
{
    const int n_it = TREE ? 8 * df_num_panels(a, j) : 8 * j;
    if (n_it == 0)
        return;   // no earlier panel: the workers hold all their tiles from the start
    const int fr = lane & 15, fk = lane >> 4;
    const int K0 = j * kNB, R0 = R * kNB;
    double4_t acc[13];
    for_tiles([&](auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I >= kHelpSplit)
            load_tile<WK, I, HAS_T>(acc, a.S, a.ld, a.n_pad, K0, R0, fr, fk);
        else
            acc[I] = (double4_t){ 0.0, 0.0, 0.0, 0.0 };
    }, Seq13{});
    for (int it = 0; it + 1 < n_it; ++it) {
        const double* XJ = m.Xs + (it & 1) * 2 * kDfXs;
        const double* XR = XJ + kDfXs;
        __syncthreads();
        if (WK == 0 || HAS_T) {
            Ops o[2];
            load_ops_slice<WK, HAS_T>(XJ, XR, fr, fk, o);
            for_tiles([&](auto idx) {
                constexpr int I = decltype(idx)::value;
                if constexpr ((HAS_T || !is_t(WK, I)) && I >= kHelpSplit)
                    mfma_tile<WK, I>(acc, o);
            }, Seq13{});
        }
    }
    double* M = m.RA + WK * (13 - kHelpSplit) * 256;
    for_tiles([&](auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I >= kHelpSplit) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                M[((I - kHelpSplit) * 4 + r) * 64 + lane] = acc[I][r];
        }
    }, Seq13{});
    __syncthreads();   // (the workers read behind this barrier; a wave that has ended no longer counts for the later ones)
}

// A give-up anywhere is a synchronisation failure, not an indefinite matrix: the pass pauses (LmCtl::done = 2) and the host
// redoes the factorisation without the dataflow.  EVERY workgroup reports for itself -- the one that gave up, and any that
// ends after somebody raised the abort word.  (Until round 3 only the last block column's workgroup did, on the grounds that
// it ends after everybody else; with a tree ordering of a kept family whose co-observation graph is not connected that is
// not true -- the last column depends on its own component only -- and a give-up in the other component went unreported.)
__device__ __forceinline__ void report_give_up(const DfArgs& a, const unsigned epoch, const int& s_timeout)
{
    if (threadIdx.x == 0
        && (s_timeout || __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch))
        raise_sync_timeout(a.ctl, 1);
}

template <bool HAS_T, bool TREE, int MODE>
__device__ __forceinline__ void role(const DfArgs& a, const int j, const int R, double* smem)
{
    constexpr bool BULK = (MODE & 1) != 0, HELP = (MODE & 2) != 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K0 = j * kNB;
    const int R0 = R * kNB;
    const int n_blk = a.n_blk, ld = a.ld, n_pad = a.n_pad;
    if (TREE && HAS_T && R < n_blk && !nz_bit(a.nz, R, j))
        return;   // L(R, j) is structurally zero: nothing to compute, nothing to publish (its consumers know)
    const unsigned epoch = *a.epoch_word + 1u;
    Lds m;
    m.RA = smem;
    m.Pd = m.RA + 64 * kLdT;
    m.Pt = m.Pd + 2 * 64 * kPsD;
    m.invd = m.Pt + 2 * 64 * kPsD;
    m.Pb = m.invd + 64;
    m.Nd = m.Pb + 8 * kPsD;
    m.Xs = m.Nd + 8 * kPsD;
    m.stamp_j = -1;
#ifdef VMM_STAMPS
    if (TREE ? !HAS_T : (HAS_T && R == j + 1))   // tree orderings: the diagonal-only workgroup (block (j+1, j) may be empty)
        m.stamp_j = j;
    {
        const bool stamp_on = m.stamp_j >= 0 && w == 0;
        const int stamp_j = m.stamp_j;
        DF_RT(0);
    }
#endif
    double* Li = m.Pd;                       // diagonal factor, row stride kLd, for the block inverse: over the panel
    double* di = Li + 64 * kLd;              // and slice buffers (Pd, Pt, invd, Xs), dead after the last round
    __shared__ int s_timeout;
    if (tid == 0)
        s_timeout = 0;
    SliceMap sm;
    sm.G = a.G;
    sm.n_blk = n_blk;
    sm.slot = TREE ? a.slot : nullptr;
    __syncthreads();   // s_timeout
    bool ok = true;
    if (w < 2)
        pivot_path<HAS_T, TREE, MODE>(a, w, lane, j, R, m, sm, epoch, &s_timeout, ok);
    else if (w == 2)
        worker_path<0, HAS_T, TREE, HELP>(a, lane, j, R, m);
    else if (w == 3)
        worker_path<1, HAS_T, TREE, HELP>(a, lane, j, R, m);
    else {   // HELP only (six waves)
        if (w == 4)
            helper_path<0, HAS_T, TREE>(a, lane, j, R, m);
        else
            helper_path<1, HAS_T, TREE>(a, lane, j, R, m);
        return;
    }
    __syncthreads();   // the pivot waves store their rows of the result tile behind the last round's barrier
    // results for the kernels after this launch
    if (!HAS_T) {
        const double iv = tid < 64 ? m.invd[tid] : 0.0;   // invd is about to be overwritten by Li
        __syncthreads();
        if (tid < 64)
            a.dinv[K0 + tid] = iv;
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const double v = (c <= r) ? m.RA[c * kLdT + r] : 0.0;
            if (c <= r)
                a.Ld[(int64_t)j * 4096 + r * 64 + c] = v;
            Li[r * kLd + c] = v;
        }
        if (tid < 64)
            di[tid] = iv;
        __syncthreads();
        if (j < n_blk - 1)   // the chain solves the last block directly
            chol_inverse_lds(Li, di, a.Linv + (int64_t)j * 4096);
        report_give_up(a, epoch, s_timeout);
        return;
    }
    for (int idx = tid; idx < 64 * 32; idx += 256) {
        const int rr = idx >> 5, c = (idx & 31) * 2;
        if (R0 + rr <= n_pad)
            *reinterpret_cast<double2*>(a.S + (int64_t)(R0 + rr) * ld + K0 + c)
                = make_double2(m.RA[rr * kLd + c], m.RA[rr * kLd + c + 1]);
    }
    if (BULK) {
        // the block once more for the workgroups that get to this panel when it is long complete: [column][row], what a
        // consumer stages slice by slice (the granules carried the same values), then the completion word -- every thread's
        // stores made visible (release at agent scope), then one thread says so
        double* cb = a.Gc + sm.index(j, R) * 4096;
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int c = idx >> 6, rr = idx & 63;
            cb[idx] = m.RA[rr * kLd + c];
        }
        __threadfence();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(a.done + sm.index(j, R), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (j == n_blk - 1 && tid == 0 && !ok)
        a.ctl->lin_fail = 1;   // (with a give-up the result is NaN-poisoned and `ok` says nothing: the pass is redone anyway)
    report_give_up(a, epoch, s_timeout);
}

} // namespace df2

template <bool TREE, int MODE>
__device__ __forceinline__ void chol_dataflow_body(DfArgs& a)
{
    if (a.ctl->done)
        return;
    a.spin_limit = (a.ctl->spin_limit_df && (a.ctl->spin_wg < 0 || a.ctl->spin_wg == (int)blockIdx.x)) ? a.ctl->spin_limit_df
                                                                                                          : kDfSpinDefault;
    phase_stamp(a.ctl, 3);
    if (a.ctl->lin_fail)
        return;
    __shared__ __attribute__((aligned(16))) double smem[kDfSmem];
    if (TREE) {
        // only the non-zero blocks of the factor have a workgroup (listed panel-major by the host)
        const int j = a.wg[2 * (int)blockIdx.x], R = a.wg[2 * (int)blockIdx.x + 1];
        if (R > j)
            df2::role<true, TREE, MODE>(a, j, R, smem);
        else
            df2::role<false, TREE, MODE>(a, j, j, smem);
        return;
    }
    int b = (int)blockIdx.x, j = 0;
    for (; j < a.n_blk; ++j) {
        const int cnt = a.n_blk - j + 1;
        if (b < cnt)
            break;
        b -= cnt;
    }
    if (j >= a.n_blk)
        return;
    if (b < a.n_blk - j)
        df2::role<true, TREE, MODE>(a, j, j + 1 + b, smem);
    else
        df2::role<false, TREE, MODE>(a, j, j, smem);
}

__global__ __launch_bounds__(256) void k_chol_dataflow(DfArgs a)
{
    chol_dataflow_body<false, 0>(a);
}

// the same launch with the compact-copy path of pivot_path (BULK) for dense systems: workgroups that are dispatched late (22 to
// 48 block columns, the 34-column tail of a large system) read the panels that are complete by then from their compact
// copies.  Not faster there (launch_dataflow), kept as the tested dense form of what the tree-ordered kernel uses
__global__ __launch_bounds__(256) void k_chol_dataflow_bulk(DfArgs a)
{
    chol_dataflow_body<false, 1>(a);
}

// the same launch for a factor with a block structure (DfArgs::nz / order: tree orderings of the kept family)
__global__ __launch_bounds__(256) void k_chol_dataflow_tree(DfArgs a)
{
    chol_dataflow_body<true, 1>(a);
}

// six waves per workgroup: two helper workers while earlier panels are applied (pivot_path, HELP)
__global__ __launch_bounds__(384) void k_chol_dataflow_tree_help(DfArgs a)
{
    chol_dataflow_body<true, 3>(a);
}

// One launch per block column k: workgroups [0, n_panel) factor panel k (with the lazy update of their own column
// from the pending panels k-1 and, for even k of the paired launches, k-2), the others apply a trailing update.  While
// the update is what a launch waits for (more than kPairMinBlocks block columns left) it is a rank-128 one: launches 2m
// and 2m+1 share the update of the panels 2m-2 and 2m-1 (PA, PB) on the block columns >= c0 = 2m+1 (tiles [t0, t1) of
// pair_tile_index each; launch 2m takes block column 2m+1, which the next panel needs, and about half of the rest).
// Near the end a launch is as long as its panel chain and the second lazy panel of the paired form (+6 us on every other
// launch) costs more than the saved traffic: PB == nullptr = the rank-64 update of panel k-1 (PA) on the columns >= k+1.
// The two parts of a launch touch disjoint tiles and both only need results of earlier launches, so the update
// (throughput work) runs beside the latency-bound panel instead of in front of it.
__global__ __launch_bounds__(256) void k_chol_step(LmCtl* ctl, double* __restrict__ S, int ld, int n_pad, int n_blk,
                                                   int k, int n_panel, double* __restrict__ Pcur,
                                                   const double* __restrict__ Pprev, const double* __restrict__ Pprev2,
                                                   double* __restrict__ dinv, double* __restrict__ Ld,
                                                   double* __restrict__ Linv, const double* __restrict__ PA,
                                                   const double* __restrict__ PB, int c0, int t0, int t1, int n_upd_wg,
                                                   unsigned* tile_ctr)
{
    if (ctl->done)
        return;
    if (k == 0)
        phase_stamp(ctl, 3);
    if (ctl->lin_fail)
        return;
    // the tile counter of launch k is word k & 1; launch k resets the other word for launch k + 1 (launches 0 and 1
    // have no trailing update: whatever an earlier factorisation left behind, word k & 1 is zero at launch k)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        tile_ctr[(k + 1) & 1] = 0u;
    __shared__ __attribute__((aligned(16))) double smem[kStepSmem];
    if ((int)blockIdx.x < n_panel) {
        chol_panel_wg(ctl, S, ld, n_pad, k, Pcur, Pprev, Pprev2, dinv, Ld, smem);
        if (t1 > t0) {   // the panel is stored: help with the trailing update
            __syncthreads();
            if (PB)
                chol_update2_wg(S, ld, n_blk, c0, tile_ctr + (k & 1), t0, t1, PA, PB, smem);
            else
                chol_update_wg(S, ld, n_blk, c0 - 2, tile_ctr + (k & 1), t1, PA, smem);
        }
    } else if ((int)blockIdx.x < n_panel + n_upd_wg) {
        if (PB)
            chol_update2_wg(S, ld, n_blk, c0, tile_ctr + (k & 1), t0, t1, PA, PB, smem);
        else   // rank-64 update of the single panel PA on the block columns >= c0 (tiles [0, t1))
            chol_update_wg(S, ld, n_blk, c0 - 2, tile_ctr + (k & 1), t1, PA, smem);
    } else   // last workgroup of launches k >= 1: invert the diagonal factor of block k-1
        chol_inverse_wg(Ld + (int64_t)(k - 1) * 4096, dinv + (k - 1) * kNB, Linv + (int64_t)(k - 1) * 4096, smem);
}

// The look-ahead launches leave the last diagonal block uninverted (the chain solves it directly); the
// covariance forward substitution needs all of them.
__global__ __launch_bounds__(256) void k_chol_inverse(const LmCtl* ctl, const double* __restrict__ Ld,
                                                      const double* __restrict__ dinv, double* __restrict__ Linv, int k)
{
    if (ctl->done || ctl->lin_fail)
        return;
    __shared__ __attribute__((aligned(16))) double smem[64 * kLd + 64];
    chol_inverse_wg(Ld + (int64_t)k * 4096, dinv + k * kNB, Linv + (int64_t)k * 4096, smem);
}

void launch_chol_inverse(Engine& e, int k)
{
    hipLaunchKernelGGL(k_chol_inverse, dim3(1), dim3(256), 0, e.stream, (const LmCtl*)e.ctl, (const double*)e.Ldiag,
                       (const double*)e.dinv, e.Linv, k);
}

constexpr int kPairMinBlocks = 46;   // block columns left below which the launches stop pairing their trailing updates

static int update_tiles(int n_blk, int k)   // tiles of the trailing update of panel k: columns >= k+2
{
    int tiles = 0;
    for (int r = k + 2; r <= n_blk; ++r)
        tiles += ((r < n_blk) ? r : n_blk - 1) - (k + 1);
    return tiles;
}

int dataflow_workgroups(int n_blk) { return n_blk * (n_blk + 1) / 2 + n_blk; }

// How many workgroups the one-launch factorisation may have.  Up to 48 block columns (1224 workgroups) -- more than the
// chip holds at one per CU.  Workgroups are panel-major; a tile only waits for lower-numbered workgroups and for the
// diagonal workgroup of its own column (fewer than n_blk + 1 <= n_cu numbers ahead), so with the in-order dispatch of
// the hardware the lowest unfinished workgroup always has its producers resident or finished and the launch drains with
// only a prefix resident.  HIP does not promise that order: the bounded spins and the redo of a pass that gave up
// waiting (k_chol_step) make a wrong guess slow, not wrong.  Measured (MI355X, us per factorisation, this kernel vs one
// k_chol_step launch per column): 24 blocks 328 vs 584, 30: 461 vs 744, 38: 701 vs 975, 47: 1097 vs 1232,
// 60: 1948 vs 1711, 94: 6254 vs 3713 -- hence 48.  VMM_BA_DF_MAX_WG overrides the limit (experiments).
int dataflow_max_workgroups(int n_cu)
{
    static const int env = [] {
        const char* v = getenv("VMM_BA_DF_MAX_WG");
        return v ? atoi(v) : 0;
    }();
    if (env > 0)
        return env;
    const int kMaxBlocks = 48;
    return n_cu > kMaxBlocks ? std::max(n_cu, dataflow_workgroups(kMaxBlocks)) : n_cu;
}

// How many of the LAST block columns of an n_blk-column system the one-launch kernel factors: all of them when it may
// (dataflow_max_workgroups), else a tail -- the launches of k_chol_step near the end are bound by their panel chain
// (~25-31 us per block column at n = 6000, whatever the trailing update costs) while the dataflow kernel needs ~14 us
// per column at 24 to 38 columns.  VMM_BA_CHOL_TAIL sets the tail length (0: none).
int dataflow_blocks(int n_blk, int n_cu)
{
    if (dataflow_workgroups(n_blk) <= dataflow_max_workgroups(n_cu))
        return n_blk;
    static const int env = [] {
        const char* v = getenv("VMM_BA_CHOL_TAIL");
        return v ? atoi(v) : -1;
    }();
    int tail = env >= 0 ? env : 34;
    if (dataflow_workgroups(tail) > dataflow_max_workgroups(n_cu) || n_blk > n_cu)
        return 0;
    tail = std::min(tail, n_blk - 2);
    return tail - ((n_blk - tail) & 1);   // the step launches come in pairs: an even number of them in front
}

// The k_chol_step launches that factor the leading n_blk - n_df block columns (and, when a dataflow tail follows, hand
// the rest of the matrix over with every update applied).  Pure host logic, also exported for the schedule test
// (vmm_ba_debug_chol_schedule): tests/test_host_cpu.py replays it for every size and checks that each tile receives
// each panel exactly once, from a panel of an earlier launch, before its block column is factored.
//   launches k < k_pair are paired (rank-128 updates: launches 2m and 2m+1 share the pair of panels 2m-2, 2m-1 on the
//   block columns >= 2m+1), launch k_pair finishes the last pair alone, later ones are single (rank-64: panel k-1 on the
//   columns >= k+1): measured at n = 6000, the pair wins while more than ~46 block columns are left.
std::vector<CholLaunch> chol_step_schedule(int n_blk, int n_df)
{
    std::vector<CholLaunch> out;
    const int n_step = n_blk - n_df;   // even when a tail follows (dataflow_blocks)
    int k_pair = 0;
    while (n_blk - k_pair > kPairMinBlocks)
        k_pair += 2;
    if (k_pair > n_step)
        k_pair = n_step;       // the hand-over launch then finishes the last pair
    for (int k = 0; k < n_step; ++k) {
        CholLaunch L = { k, { -1, k > 0 ? k - 1 : -1 }, { -1, -1 }, k + 1, 0, 0 };
        if (k >= 2 && k <= k_pair) {
            const int m = k / 2;
            L.c0 = 2 * m + 1;
            L.upd[0] = 2 * m - 2;
            L.upd[1] = 2 * m - 1;
            if (!(k & 1))
                L.lazy[0] = k - 2;
            if (L.c0 <= n_blk - 1) {
                const int n_first = n_blk - L.c0 + 1;
                const int total = n_first + update_tiles(n_blk, L.c0 - 1);
                // launch 2m takes block column 2m+1 (the next panel needs it) and about half of the rest
                const int half = k == k_pair ? total : std::max(n_first, (total + 1) / 2);
                L.t0 = (k & 1) ? half : 0;
                L.t1 = (k & 1) ? total : half;
            }
        } else if (k >= 1 && k > k_pair) {
            L.upd[0] = k - 1;
            L.t1 = update_tiles(n_blk, k - 1);
        }
        out.push_back(L);
    }
    if (n_df > 0) {
        // hand-over to the one-launch kernel: what is still pending on every block column >= n_step (the pair of panels
        // n_step-2, n_step-1 when the last launch was a paired one, else panel n_step-1) in one update-only launch (it
        // also inverts diagonal block n_step-1)
        const bool pair = n_step <= k_pair;
        CholLaunch L = { -1, { -1, -1 }, { n_step - (pair ? 2 : 1), pair ? n_step - 1 : -1 }, n_step, 0, 0 };
        L.t1 = pair ? (n_blk - L.c0 + 1) + update_tiles(n_blk, L.c0 - 1) : update_tiles(n_blk, L.c0 - 2);
        out.push_back(L);
    }
    return out;
}

// tile t of a launch's update list (host copy of what the kernel computes)
void chol_schedule_tile(int n_blk, const CholLaunch& L, int t, int* bi, int* bj)
{
    if (L.upd[1] >= 0)
        pair_tile_index(n_blk, L.c0, t, *bi, *bj);
    else
        update_tile_index(n_blk, L.c0 - 2, t, *bi, *bj);
}

static void launch_dataflow(Engine& e, double* S, int n_pad, int ld, LmCtl* ctl, int first_blk, int n_blk)
{
    DfArgs a;
    a.ctl = ctl;
    a.S = S + (int64_t)first_blk * kNB * (ld + 1);
    a.ld = ld;
    a.n_pad = n_pad - first_blk * kNB;
    a.n_blk = n_blk - first_blk;
    a.dinv = e.dinv + first_blk * kNB;
    a.Ld = e.Ldiag + (int64_t)first_blk * 4096;
    a.Linv = e.Linv + (int64_t)first_blk * 4096;
    a.G = e.df_gran;
    a.epoch_word = e.flags + 256;
    a.abort_word = e.flags + 257;
    a.spin_limit = 0;
    a.nz = (first_blk == 0 && e.chol_nz_on) ? e.chol_nz : nullptr;
    a.order = a.nz ? e.chol_order : nullptr;
    a.wg = a.nz ? e.df_wg : nullptr;
    a.slot = a.nz ? e.df_slot : nullptr;
    a.Gc = e.df_compact;
    a.done = e.df_done;
    const char* const bulk_v = getenv("VMM_BA_DF_BULK");   // (read per launch: tests switch it within one process)
    const int bulk_env = bulk_v ? atoi(bulk_v) : -1;
    // Dense systems: measured (MI355X, us per factorisation, granules only / compact copies): 24 block columns 278 / 295,
    // 30: 387 / 390, 38: 582 / 583, 47: 896 / 891, the 34-column tail at n = 6000: 3023 / 3041 -- a late workgroup there is
    // bound by its two worker waves (26 MFMAs per slice each), not by its sweeps; so only on request (VMM_BA_DF_BULK=1, tested).
    // Tree orderings (k_chol_dataflow_tree) always: 2000 x 1000 close-up 1004 -> 874 us, 500 x 200 close-up 205 -> 200.
    const bool bulk = e.df_compact && e.df_done && bulk_env > 0;
    // Helper waves (six waves per workgroup, the same bits): measured (MI355X, factorisation + solve, four / six waves) --
    // tree orderings: 2000 x 1000 close-up (109 block columns) 876 / 810 us, 500 x 200 close-up (22) 202 / 202, corridor
    // 120 / 124; dense: 19 block columns 226 / 238, 24: 281 / 294, 30: 389 / 410, 38: 582 / 617, 47: 892 / 954, the 34-column
    // tail at n = 6000 3042 / 3070.  So: large tree-ordered factors only (VMM_BA_DF_HELP=0 / 1 decides otherwise; the dense
    // kernels were measured with an instantiation that is not kept).
    const char* const help_v = getenv("VMM_BA_DF_HELP");
    const bool help = help_v ? help_v[0] == '1' : a.n_blk >= 64;
    if (a.nz && help)
        hipLaunchKernelGGL(k_chol_dataflow_tree_help, dim3(e.n_df_wg), dim3(384), 0, e.stream, a);
    else if (a.nz)
        hipLaunchKernelGGL(k_chol_dataflow_tree, dim3(e.n_df_wg), dim3(256), 0, e.stream, a);
    else if (bulk)
        hipLaunchKernelGGL(k_chol_dataflow_bulk, dim3(dataflow_workgroups(a.n_blk)), dim3(256), 0, e.stream, a);
    else
        hipLaunchKernelGGL(k_chol_dataflow, dim3(dataflow_workgroups(a.n_blk)), dim3(256), 0, e.stream, a);
}

static void launch_backsolve_chain(Engine& e, double* S, int n_pad, int ld, double* y, LmCtl* ctl)
{
    const int n_blk = n_pad / kNB;
    if (e.chol_nz_on)
        hipLaunchKernelGGL(k_backsolve_chain_tree, dim3(backsolve_chain_workgroups(n_blk)), dim3(256), 0, e.stream, ctl, S, ld,
                           n_pad, n_blk, y, e.dinv, e.gran, e.flags + 256, (const double*)e.Ldiag, (const double*)e.Linv,
                           (const unsigned long long*)e.chol_nz, e.flags + 261);
    else
        hipLaunchKernelGGL(k_backsolve_chain, dim3(backsolve_chain_workgroups(n_blk)), dim3(256), 0, e.stream, ctl, S, ld,
                           n_pad, n_blk, y, e.dinv, e.gran, e.flags + 256, (const double*)e.Ldiag, (const double*)e.Linv);
}

void launch_cholesky_solve(Engine& e, double* S, int n_pad, int ld, double* y, LmCtl* ctl, bool safe)
{
    const int n_blk = n_pad / kNB;
    const bool chain = n_blk <= e.n_cu && e.flags && e.gran && !e.no_chain && !safe;
    // a tree-ordered handle: the one-launch kernel whatever the size (only the non-zero blocks have workgroups)
    const int n_df = (chain && e.df_gran && !e.no_dataflow) ? (e.chol_nz_on ? n_blk : dataflow_blocks(n_blk, e.n_cu)) : 0;
    if (n_df == n_blk) {
        // one launch for the factorisation + forward substitution, one for the back-substitution chain (which
        // bumps the epoch both kernels tag their granules with)
        launch_dataflow(e, S, n_pad, ld, ctl, 0, n_blk);
        launch_backsolve_chain(e, S, n_pad, ld, y, ctl);
        return;
    }
    for (const CholLaunch& L : chol_step_schedule(n_blk, n_df)) {
        const int k = L.k >= 0 ? L.k : n_blk - n_df;   // (the hand-over launch carries the number of the first tail column)
        int n_panel = 0;
        if (L.k >= 0) {
            const int rows_below = n_pad + 1 - (k + 1) * kNB;
            n_panel = 1 + (rows_below + 63) / 64;
        }
        auto panel = [&](int p) { return p >= 0 ? (const double*)e.P4[p & 3] : (const double*)nullptr; };
        const int n_upd = L.t1 - L.t0;
        // all workgroups of a launch resident at once (one per CU: 160 KB of LDS): the update workgroups
        // share the CUs the panel leaves free and loop over the tiles
        const int n_upd_wg = L.k >= 0 ? std::min(n_upd, std::max(e.n_cu - n_panel - 1, e.n_cu / 4))
                                      : std::min(n_upd, e.n_cu - 1);
        const int grid = n_panel + n_upd_wg + (k > 0 ? 1 : 0);
        hipLaunchKernelGGL(k_chol_step, dim3(grid), dim3(256), 0, e.stream, ctl, S, ld, n_pad, n_blk, k, n_panel,
                           L.k >= 0 ? e.P4[k & 3] : (double*)nullptr, panel(L.lazy[1]), panel(L.lazy[0]), e.dinv, e.Ldiag,
                           e.Linv, panel(L.upd[0]), panel(L.upd[1]), L.c0, L.t0, L.t1, n_upd_wg, e.flags + 258);
        if (getenv("VMM_BA_DEBUG")) {
            const hipError_t le = hipPeekAtLastError();
            if (le != hipSuccess)
                fprintf(stderr, "[vmm_ba debug] k_chol_step k=%d grid=%d: %s\n", k, grid, hipGetErrorString(le));
        }
    }
    if (n_df > 0)   // the trailing n_df x n_df blocks (+ right-hand side row) in one launch
        launch_dataflow(e, S, n_pad, ld, ctl, n_blk - n_df, n_blk);
    // one chained launch while every workgroup of the chain is certainly resident (one per CU); the per-block
    // kernels otherwise
    if (chain) {
        launch_backsolve_chain(e, S, n_pad, ld, y, ctl);
    } else {
        for (int kb = n_blk - 1; kb >= 0; --kb)
            hipLaunchKernelGGL(k_backsolve_step, dim3(kb + 1), dim3(256), 0, e.stream, ctl, S, ld, n_pad, n_blk, kb, y,
                               e.dinv, (const double*)e.Ldiag);
    }
}

// Touches every kernel of this file once (vmm_ba_create): the code object is loaded and the kernel's resources
// are known before any launch is recorded into a hipGraph (nothing may be loaded lazily under stream capture).
int preload_chol_kernels()
{
    hipFuncAttributes at;
    int bad = 0;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsolve_step)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsolve_chain)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsolve_chain_tree)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_dataflow)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_dataflow_bulk)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_dataflow_tree)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_dataflow_tree_help)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_step)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_chol_inverse)) != hipSuccess;
    return bad;
}

} // namespace vmm
