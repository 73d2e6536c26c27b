// Pivot-block Cholesky and row scaling with one matrix ROW per lane (gfx950).
//
// The pivot chain of the blocked Cholesky (kernels_chol.hip) used to factor its 8x8 pivot block redundantly in every
// lane: ~176 f64 instructions per block, and on this hardware that code is ISSUE-bound (a f64 VALU instruction occupies
// the SIMD for ~6 cycles whether or not it depends on the previous one -- tools/micro/lat2.hip -- so the 1100 cycles of
// the 8x8 block were its instruction count, not its dependent chain of ~400).  Here lane i of every 16-lane DPP row owns
// row i of a W x W pivot block (W = 8 or 16): the trailing update of step j is ONE instruction per remaining column,
//     v_fmac_f64_dpp  a[c], -a[j] (row_newbcast:c), a[j]        a[i][c] -= L[c][j] * L[i][j]
// whose first factor is read from lane c of the DPP row by the operand fetch itself (gfx90a+: 64-bit DPP exists only with
// row_newbcast, and this is what it is for).  A 16x16 block costs 264 instructions instead of ~816, the four DPP rows of
// a wave hold four identical copies, and the scaling of a panel row x <- x L^-T takes its L entries the same way.
//
// Hazard: a DPP operand written by the previous VALU instruction needs two wait states and the compiler does not look
// inside inline assembly, so every DPP statement starts with s_nop 1.
#pragma once

namespace vmm {
namespace dpp {

// v of lane N (0..15) of this lane's 16-lane row
template <int N>
__device__ __forceinline__ double bcast(const double v)
{
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
    return r;
}

// acc -= (b of lane N of this row) * own, one rounding
template <int N>
__device__ __forceinline__ void fnma_bcast(double& acc, const double b, const double own)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(b), "v"(own), "n"(N));
}

// 1/sqrt(v): v_rsq_f64 seed + one third-order correction (~1 ulp), branch-free; a non-positive or non-finite pivot gives
// NaN and clears ok (same contract as safe_rsqrt of potrf64.inc)
__device__ __forceinline__ double rsqrt_refined(const double v, bool& ok)
{
    ok = ok && (v > 0.0) && isfinite(v);
    const double y0 = __builtin_amdgcn_rsq(v);
    const double e = fma(-v * y0, y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

template <int W, int J, int C>
struct UpdateCols {
    static __device__ __forceinline__ void run(double (&a)[W])
    {
        if constexpr (C < W) {
            fnma_bcast<C>(a[C], a[J], a[J]);
            UpdateCols<W, J, C + 1>::run(a);
        }
    }
};

template <int W, int J>
struct CholSteps {
    static __device__ __forceinline__ void run(double (&a)[W], double (&inv)[W], bool& ok)
    {
        if constexpr (J < W) {
            const double t = bcast<J>(a[J]);          // the pivot A[J][J] after J updates
            const double r = rsqrt_refined(t, ok);
            inv[J] = r;
            a[J] *= r;                                 // L[i][J] (lane J: sqrt(t))
            UpdateCols<W, J, J + 1>::run(a);           // right-looking: the next pivot is one instruction behind
            CholSteps<W, J + 1>::run(a, inv, ok);
        }
    }
};

// In: a[c] = A[i][c] for c <= i, i = lane & 15 < W (entries c > i and lanes >= W: anything finite or not, never read by a
// valid result).  Out: a[c] = L[i][c] (c <= i), inv[j] = 1 / L[j][j] in every lane, ok = all pivots positive and finite.
template <int W>
__device__ __forceinline__ void chol_rows(double (&a)[W], double (&inv)[W], bool& ok)
{
    static_assert(W == 8 || W == 16, "one DPP row holds at most 16 matrix rows");
    CholSteps<W, 0>::run(a, inv, ok);
}

template <int W, int Q, int C>
struct ScaleCols {
    static __device__ __forceinline__ void run(double (&x)[W], const double (&l)[W])
    {
        if constexpr (C < W) {
            fnma_bcast<C>(x[C], l[Q], x[Q]);           // x[C] -= L[C][Q] * x[Q]
            ScaleCols<W, Q, C + 1>::run(x, l);
        }
    }
};

template <int W, int Q>
struct ScaleSteps {
    static __device__ __forceinline__ void run(double (&x)[W], const double (&l)[W], const double (&inv)[W])
    {
        if constexpr (Q < W) {
            x[Q] *= inv[Q];
            ScaleCols<W, Q, Q + 1>::run(x, l);
            ScaleSteps<W, Q + 1>::run(x, l, inv);
        }
    }
};

// x <- x L^-T for this lane's own row x[0..W-1] (any lane: lane = a row of the panel); l / inv as chol_rows left them in
// the lanes of this lane's DPP row
template <int W>
__device__ __forceinline__ void scale_row(double (&x)[W], const double (&l)[W], const double (&inv)[W])
{
    ScaleSteps<W, 0>::run(x, l, inv);
}

} // namespace dpp
} // namespace vmm
