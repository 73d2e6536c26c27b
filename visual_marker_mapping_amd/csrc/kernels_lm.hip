// Device-resident trust-region control (gfx950): the policy of Ceres' TrustRegionMinimizer +
// LevenbergMarquardtStrategy that the reference runs with default options through ceres::Solve
// (src/TagReconstructor.cpp:725-738), restated per SURVEY.md Appendix A.4.  All scalars of the loop
// live in one LmCtl block in HBM; every kernel of an iteration starts by reading its `done` flag, so
// the host can enqueue iterations ahead and only polls.
#include <float.h>

#include "engine.hpp"

namespace vmm {

// The control kernel's reduction: seven sums and two maxima over the block, needed by thread 0 only.
// One shared butterfly per wave (the number of live values halves while the span doubles: 10 + 6 lane exchanges
// instead of 9 x 6), wave totals to LDS, ONE barrier, and thread 0 adds the wave totals in wave order; nobody else
// reads them (sixteen waves each reading all of them was 4 us of LDS instructions on one compute unit).
// v[0..6] sums, v[7] must be 0, m[0..1] maxima (>= 0).  sh: [waves][10].  Result valid in thread 0.  Deterministic.
__device__ __forceinline__ void control_reduce(double (&v)[8], double (&m)[2], double* sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const int off = 32 >> level;
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < (4 >> level); ++j) {
            const double keep = hi ? v[2 * j + 1] : v[2 * j];
            const double send = hi ? v[2 * j] : v[2 * j + 1];
            v[j] = keep + __shfl_xor(send, off, 64);
        }
    }
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1)
        v[0] += __shfl_xor(v[0], off, 64);
    {
        const bool hi = (lane & 32) != 0;
        const double keep = hi ? m[1] : m[0];
        const double send = hi ? m[0] : m[1];
        m[0] = fmax(keep, __shfl_xor(send, 32, 64));
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1)
            m[0] = fmax(m[0], __shfl_xor(m[0], off, 64));
    }
    // value index a lane ends with: bit 0 from lane bit 5, bit 1 from lane bit 4, bit 2 from lane bit 3
    if ((lane & 7) == 0)
        sh[wave * 10 + (((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2))] = v[0];
    if ((lane & 31) == 0)
        sh[wave * 10 + 8 + (lane >> 5)] = m[0];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double r = sh[k];
            for (int w = 1; w < nw; ++w)
                r += sh[w * 10 + k];
            v[k] = r;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double r = sh[8 + k];
            for (int w = 1; w < nw; ++w)
                r = fmax(r, sh[w * 10 + 8 + k]);
            m[k] = r;
        }
    }
}

struct PoseViews {
    int n_cams, n_tags;
    int tag_euclid;   // point landmarks: a "tag" slot is a pair of 3-D points, Plus is plain addition of six numbers
    double* cam_qt;
    double* tag_qt;
    double* cam_cand;
    double* tag_cand;
};

// Plus of block p: cameras (and tag poses) = translation + QuaternionParameterization; point pairs = addition
__device__ __forceinline__ void block_plus(const PoseViews& v, int p, const double* __restrict__ x, const double d[6],
                                           double out[7])
{
    if (v.tag_euclid && p >= v.n_cams) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            out[k] = x[k] + d[k];
        out[6] = x[6];
    } else {
        pose_plus(x, d, out);
    }
}

__device__ __forceinline__ double* pose_ptr(const PoseViews& v, int p, bool cand)
{
    if (p < v.n_cams)
        return (cand ? v.cam_cand : v.cam_qt) + 7 * (int64_t)p;
    return (cand ? v.tag_cand : v.tag_qt) + 7 * (int64_t)(p - v.n_cams);
}

// The control kernel of an LM iteration, one block, between the evaluation at the candidate and the elimination:
//  (0) the decision on the step whose candidate was just evaluated -- ComputeTrustRegionStep (model cost),
//      ParameterToleranceReached, FunctionToleranceReached, IsStepSuccessful, HandleSuccessfulStep /
//      HandleUnsuccessfulStep / HandleInvalidStep, LevenbergMarquardtStrategy::StepAccepted / StepRejected; an
//      accepted candidate becomes x and its evaluation becomes the evaluation at x (w_which flips).  Iteration zero
//      has no step to judge: the evaluation was at x itself;
//  (a) if x moved (TrustRegionMinimizer::IterationZero / EvaluateGradientAndJacobian): total cost, Jacobi scaling
//      (first evaluation only), gradient max-norm |Plus(x,-g) - x|_inf, completion of the pending iteration record;
//  (b) FinalizeIterationAndCheckIfMinimizerCanContinue: push the record, termination tests;
//  (c) first half of LevenbergMarquardtStrategy::ComputeStep: the LM diagonal D^2.
// (Two kernels until the evaluation moved to the candidate put them next to each other; one launch saves the
// dependent-launch gap and a second read-modify-write of the control block.)
#ifdef VMM_STAMPS
__device__ unsigned long long g_ctl_stamps[16];
#define CTL_RT(slot)                                                    \
    do {                                                                \
        if (threadIdx.x == 0)                                           \
            g_ctl_stamps[slot] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)
extern "C" int vmm_ba_debug_read_ctl_stamps(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ctl_stamps), sizeof(unsigned long long) * 16);
}
#else
#define CTL_RT(slot)
#endif

struct DecideArgs {
    const double* pose_part;        // [n_pose][5], k_backsub / k_candidate
    const double* cross_parts;      // per-pose partials of the cross term (world > 1: all-reduced), summed here in pose order
    int n_cross;
    const double* cost_parts;       // per-pose costs of the evaluation at the candidate (world > 1: all-reduced)
    int n_cost;
    const double* pose_gm;          // one GPU: per-pose |Plus(x+, -g) - x+|_inf from k_reduce_pose; null: computed here
};

__global__ __launch_bounds__(1024) void k_control(LmCtl* ctl, PoseViews pv, const DecideArgs dz,
                                                 const double* __restrict__ src, double* __restrict__ dst,
                                                 size_t small_count, const double* __restrict__ H0,
                                                 const double* __restrict__ g0, const int64_t alt_off,
                                                 double* __restrict__ scale, int32_t* __restrict__ active,
                                                 double* __restrict__ diag, double* __restrict__ D2,
                                                 vmm_ba_iteration* __restrict__ trace)
{
    if (ctl->done)
        return;
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    CTL_RT(0);
    __shared__ double sh[256];
    __shared__ int s_stop, s_reuse, s_accept, s_moved;
    __shared__ double s_radius, s_lo, s_hi;
    const int tid = threadIdx.x;
    const int n_pose = pv.n_cams + pv.n_tags;
    const int n_tan = 6 * n_pose;
    const bool first = ctl->first_eval != 0;
    const bool jacobi = ctl->jacobi_scaling != 0;
    const int which = ctl->w_which;     // the copy of W and of the small blocks that belongs to x right now
    const bool staged = src != dst;     // world > 1: the evaluation went to the all-reduced staging copy
    // the blocks the evaluation at the candidate has just written
    const double* __restrict__ nb = staged ? src : dst + (which ? 0 : alt_off);
    const double* __restrict__ Hn = nb + (H0 - dst);
    const double* __restrict__ gn = nb + (g0 - dst);
    // One dependent chain of memory round trips instead of six: everything the decision AND the set-up at an accepted
    // candidate need is loaded and reduced before the decision is known (the gradient norm and, at iteration zero,
    // the scaling are properties of the candidate and of its evaluation alone); the control block travels to thread
    // 0's registers meanwhile and is stored once.
    __shared__ LmCtl s_c;   // thread 0's working copy (in registers the compiler spills it to scratch)
    if (tid == 0)
        s_c = *ctl;
    double gd = 0.0, quad = 0.0, sn = 0.0, xn = 0.0, bad = 0.0, cross = 0.0, ccost = 0.0, xn0 = 0.0, gm = 0.0;
    for (int p = tid; p < n_pose; p += (int)blockDim.x) {
        if (!first) {
            const double* o = dz.pose_part + 5 * (int64_t)p;
            gd += o[0];
            quad += o[1];
            sn += o[2];
            xn += o[3];
            bad = fmax(bad, o[4]);
        }
        // (a) at the candidate: Jacobi scaling and the reduced program (iteration zero only), gradient max-norm
        const double* Hp = Hn + 36 * (int64_t)p;
        int act;
        if (first) {
            // blocks with zero Jacobian columns (constant origin tag, poses without observations) are not part of
            // Ceres' reduced program
            act = (Hp[0] + Hp[7] + Hp[14]) > 0.0 ? 1 : 0;
            active[p] = act;
            for (int k = 0; k < 6; ++k)
                scale[6 * (int64_t)p + k] = jacobi ? 1.0 / (1.0 + sqrt(Hp[7 * k])) : 1.0;
        } else {
            act = active[p];
        }
        if (act) {
            const double* x = pose_ptr(pv, p, true);
            if (first)
                for (int k = 0; k < 7; ++k)
                    xn0 += x[k] * x[k];
            if (dz.pose_gm) {
                gm = fmax(gm, dz.pose_gm[p]);
            } else {
                // 700 quaternion exponentials on one compute unit cost ~8 us: only where the gradient is not
                // known before this kernel (world > 1: behind the all-reduce)
                double ng[6], xp[7];
                for (int k = 0; k < 6; ++k)
                    ng[k] = -gn[6 * (int64_t)p + k];
                block_plus(pv, p, x, ng, xp);
                for (int k = 0; k < 7; ++k)
                    gm = fmax(gm, fabs(x[k] - xp[k]));
            }
        }
    }
    // the per-pose partials of the cross term (world > 1: all-reduced) are summed here in pose order; so are, on one
    // GPU, the per-pose costs of the evaluation at the candidate (world > 1: an all-reduced scalar slot)
    if (!first)
        for (int i = tid; i < dz.n_cross; i += (int)blockDim.x)
            cross += dz.cross_parts[i];
    for (int i = tid; i < dz.n_cost; i += (int)blockDim.x)
        ccost += dz.cost_parts[i];
    CTL_RT(1);
    double red[8] = { gd, quad, sn, xn, cross, ccost, xn0, 0.0 };
    double redm[2] = { bad, gm };
    control_reduce(red, redm, sh);
    CTL_RT(2);
    if (tid == 0) {
        LmCtl& c = s_c;
        gd = red[0];
        quad = red[1];
        sn = red[2];
        xn = red[3];
        cross = red[4];
        ccost = red[5];
        xn0 = red[6];
        bad = redm[0];
        gm = redm[1];
        auto span = [](unsigned long long a, unsigned long long b) { return b > a ? b - a : 0ull; };
        int accept = 0;
        // ---- (0) the decision on the step whose candidate was evaluated ----
        if (first) {
            c.w_which = which ^ 1;   // W and the small blocks at x: what the evaluation has just written
            c.phase_ticks[0] += span(c.stamp[0], t_begin);
        } else {
            const bool lin_fail = c.lin_fail != 0 || bad != 0.0;
            // model_cost_change = -(J d)^T (r + J d / 2) = -d^T g - 1/2 d^T H d   (unscaled coordinates)
            const double mcc = lin_fail ? 0.0 : -gd - 0.5 * (quad + 2.0 * cross);
            c.model_cost_change = mcc;
            c.cur.model_cost_change = mcc;
            const bool valid = !lin_fail && (mcc > 0.0);
            c.cur.step_is_valid = valid ? 1 : 0;
            if (!valid) {
                // HandleInvalidStep
                c.num_invalid++;
                if (c.num_invalid >= c.max_invalid) {
                    c.done = 1;
                    c.termination = VMM_BA_FAILURE;
                } else {
                    c.radius = c.radius / c.decrease_factor;
                    c.decrease_factor *= 2.0;
                    c.reuse_diagonal = 1;
                    c.cur.cost = c.x_cost;
                    c.cur.step_is_successful = 0;
                }
            } else {
                c.num_invalid = 0;
                double cand = ccost;
                c.num_cost_evals++;
                if (!isfinite(cand))
                    cand = DBL_MAX;
                c.cand_cost = cand;
                const double step_norm = sqrt(sn);
                c.cur.step_norm = step_norm;
                const double x_cost = c.x_cost;
                const double cost_change = x_cost - cand;
                if (step_norm <= c.parameter_tolerance * (c.x_norm + c.parameter_tolerance)) {
                    c.done = 1;   // ParameterToleranceReached: return without pushing this record
                    c.termination = VMM_BA_CONVERGENCE;
                } else if (fabs(cost_change) <= c.function_tolerance * x_cost) {
                    c.cur.cost_change = cost_change;
                    c.done = 1;   // FunctionToleranceReached
                    c.termination = VMM_BA_CONVERGENCE;
                } else {
                    c.cur.cost_change = cost_change;
                    const double rd = (cand >= DBL_MAX) ? -DBL_MAX : cost_change / mcc;
                    c.cur.relative_decrease = rd;
                    if (rd > c.min_relative_decrease) {
                        accept = 1;
                        const double q = 2.0 * rd - 1.0;
                        double den = 1.0 - q * q * q;
                        den = den < 1.0 / 3.0 ? 1.0 / 3.0 : den;
                        double r = c.radius / den;
                        c.radius = r > c.max_radius ? c.max_radius : r;
                        c.decrease_factor = 2.0;
                        c.reuse_diagonal = 0;
                        c.w_which = which ^ 1;   // the blocks evaluated at the candidate become the blocks at x
                        c.x_norm = sqrt(xn);
                        c.cur.step_is_successful = 1;
                    } else {
                        c.cur.step_is_successful = 0;
                        c.cur.cost = cand;
                        c.radius = c.radius / c.decrease_factor;
                        c.decrease_factor *= 2.0;
                        c.reuse_diagonal = 1;
                    }
                }
            }
            // phase report.  Order of the groups: evaluation at the candidate (stamp 0) -> this kernel (5) ->
            // k_form_z (2) -> factorisation (3) -> k_backsub (4) -> next evaluation (0)
            c.phase_ticks[0] += span(c.stamp[0], t_begin);
            c.phase_ticks[1] += span(c.stamp[5], c.stamp[2]);
            c.phase_ticks[2] += span(c.stamp[2], c.stamp[3]);
            c.phase_ticks[3] += span(c.stamp[3], c.stamp[4]);
            c.phase_ticks[4] += span(c.stamp[4], c.stamp[0]);
        }
        c.stamp[5] = t_begin;
        const int moved = (first || accept) ? 1 : 0;
        // ---- (a) x moved: its cost, gradient norm, and (iteration zero) the start of the trace ----
        if (!c.done && moved) {
            const double cst = ccost;
            c.x_cost = cst;
            c.num_jac_evals++;
            if (first) {
                c.first_eval = 0;
                c.initial_cost = cst;
                c.x_norm = sqrt(xn0);
                c.cur.iteration = 0;
                c.cur.step_is_valid = 1;
                c.cur.step_is_successful = 1;
            }
            c.cur.cost = cst;
            c.cur.gradient_max_norm = gm;
            if (!isfinite(cst)) {
                c.done = 1;
                c.termination = VMM_BA_FAILURE;
            }
        }
        // ---- (b) FinalizeIterationAndCheckIfMinimizerCanContinue ----
        int stop = c.done;
        if (!stop) {
            vmm_ba_iteration cur = c.cur;
            cur.trust_region_radius = c.radius;
            if (cur.step_is_successful)
                c.num_successful++;
            else
                c.num_unsuccessful++;
            if (c.records < c.trace_capacity)
                trace[c.records] = cur;
            c.records++;
            if (cur.iteration >= c.max_num_iterations) {
                stop = 1;
                c.termination = VMM_BA_NO_CONVERGENCE;
            } else if (cur.step_is_successful && cur.gradient_max_norm <= c.gradient_tolerance) {
                stop = 1;
                c.termination = VMM_BA_CONVERGENCE;
            } else if (c.radius <= c.min_radius) {
                stop = 1;
                c.termination = VMM_BA_CONVERGENCE;
            }
            if (stop) {
                c.done = 1;
            } else {
                vmm_ba_iteration z;
                z.iteration = cur.iteration + 1;
                z.step_is_valid = 0;
                z.step_is_successful = 0;
                z.reserved = 0;
                z.cost = c.x_cost;
                z.cost_change = 0.0;
                z.gradient_max_norm = cur.gradient_max_norm;  // carried until the next successful step
                z.step_norm = 0.0;
                z.relative_decrease = 0.0;
                z.trust_region_radius = 0.0;
                z.model_cost_change = 0.0;
                c.cur = z;
                c.iteration = z.iteration;
                c.lin_fail = 0;
                c.num_lm_iterations++;
            }
        }
        s_stop = stop;
        s_accept = accept;
        s_moved = moved;
        s_reuse = c.reuse_diagonal;
        s_radius = c.radius;
        s_lo = c.min_lm_diagonal;
        s_hi = c.max_lm_diagonal;
        if (!stop)
            c.reuse_diagonal = 1;
        else
            c.phase_ticks[1] += span(t_begin, __builtin_amdgcn_s_memrealtime());
        CTL_RT(3);
        *ctl = c;
        CTL_RT(4);
    }
    __syncthreads();
    CTL_RT(5);
    if (s_accept) {
        for (int i = tid; i < 7 * pv.n_cams; i += (int)blockDim.x)
            pv.cam_qt[i] = pv.cam_cand[i];
        for (int i = tid; i < 7 * pv.n_tags; i += (int)blockDim.x)
            pv.tag_qt[i] = pv.tag_cand[i];
    }
    // world > 1: the all-reduced staging copy becomes the working copy (one GPU selects through w_which)
    if (staged && s_moved)
        for (size_t i = tid; i < small_count; i += blockDim.x)
            dst[i] = src[i];
    if (s_stop)
        return;
    // ---- (c) first half of LevenbergMarquardtStrategy::ComputeStep: the LM diagonal ----
    const double* __restrict__ Hx = s_moved ? Hn : H0 + (which ? alt_off : 0);
    const double lo = s_lo, hi = s_hi;
    for (int q = tid; q < n_tan; q += (int)blockDim.x) {
        double d;
        if (!s_reuse) {
            const int p = q / 6, k = q % 6;
            const double sc = scale[q];
            d = sc * sc * Hx[36 * (int64_t)p + 7 * k];   // squared column norm of the scaled Jacobian
            d = fmin(fmax(d, lo), hi);
            diag[q] = d;
        } else {
            d = diag[q];
        }
        const double lm = sqrt(d / s_radius);           // lm_diagonal_ = sqrt(diagonal_ / radius_)
        D2[q] = lm * lm;
    }
    CTL_RT(6);
}

// One pose's share of TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost (first half): the unscaled tangent
// step d, the candidate x+ = Plus(x, d), and the pose's terms of the model cost and of the step / parameter norms:
// pose_part[p] = { d.g, d^T H d, |x - x+|^2, |x+|^2, non-finite flag } (the norms only for active poses).
__device__ __forceinline__ void candidate_for_pose(const PoseViews& pv, const int p, const double (&d)[6],
                                                   double* __restrict__ delta, const double* __restrict__ H,
                                                   const double* __restrict__ g, const int32_t* __restrict__ active,
                                                   double* __restrict__ pose_part)
{
    double bad = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        delta[6 * (int64_t)p + k] = d[k];
        if (!isfinite(d[k]))
            bad = 1.0;
    }
    const double* x = pose_ptr(pv, p, false);
    double out[7];
    block_plus(pv, p, x, d, out);
    double* c = pose_ptr(pv, p, true);
    double sn = 0.0, xn = 0.0;
    for (int k = 0; k < 7; ++k) {
        c[k] = out[k];
        const double df = x[k] - out[k];
        sn += df * df;
        xn += out[k] * out[k];
    }
    const double* Hp = H + 36 * (int64_t)p;
    double gd = 0.0, quad = 0.0;
    for (int a = 0; a < 6; ++a) {
        gd += d[a] * g[6 * (int64_t)p + a];
        double r = 0.0;
        for (int b = 0; b < 6; ++b)
            r += Hp[6 * a + b] * d[b];
        quad += d[a] * r;
    }
    const bool act = active[p] != 0;
    double* o = pose_part + 5 * (int64_t)p;
    o[0] = gd;
    o[1] = quad;
    o[2] = act ? sn : 0.0;
    o[3] = act ? xn : 0.0;
    o[4] = bad;
}

// Back-substitution of the eliminated family: y_e = L_e^{-T} (z_e - Z_e y_f), delta_e = -s_e y_e, and
// the pose's share of the model-cost cross term sum_obs delta_e^T W_ef delta_f.  With
// W_ef = s_e^-1 L_e Z_ef s_f^-1 and delta = -s y that share is (L_e^T y_e)^T (Z_e y_f) = (z_e - a)^T a
// for a = Z_e y_f, which this kernel has anyway: no second pass over the observations and W.
// One wave per eliminated pose.
// FUSE (one GPU): the candidate of every pose is formed here too -- an eliminated pose's by lane 0 of its wave, the
// kept poses' by one thread each in the workgroups behind the eliminated family's -- instead of in k_candidate.
// SPARSE: Z is the compressed Engine::Zc (one column-major 6x6 block per observation, E order); the product with y_f gathers the
// kept poses of e's observations.
template <bool FUSE, bool SPARSE>
__global__ __launch_bounds__(256) void k_backsub(const LmCtl* ctl, int n_e, int e_off_pose,
                                                 const int32_t* __restrict__ pose_task,
                                                 const double* __restrict__ Z, int ldz, int n_red,
                                                 const double* __restrict__ yf, const double* __restrict__ Le,
                                                 const double* __restrict__ ze, const double* __restrict__ scale,
                                                 double* __restrict__ step_comm, double* __restrict__ part_cross,
                                                 PoseViews pv, int f_off_pose, int n_f, int nb_e,
                                                 double* __restrict__ delta, const double* __restrict__ H0,
                                                 const double* __restrict__ g0, const int64_t alt_off,
                                                 const int32_t* __restrict__ active,
                                                 double* __restrict__ pose_part,
                                                 const int32_t* __restrict__ e_start,
                                                 const int32_t* __restrict__ e_other,
                                                 const int32_t* __restrict__ row_of)
{
    // world > 1: this rank's vote on "the factorisation of this pass gave up waiting" rides on the step all-reduce
    // (slot 7 n_e, behind the steps and the per-pose cross terms), so that all ranks pause in the same pass (k_candidate)
    if (!FUSE && blockIdx.x == 0 && threadIdx.x == 0)
        step_comm[7 * (int64_t)n_e] = (ctl->done == 2) ? 1.0 : 0.0;
    if (ctl->done)
        return;
    phase_stamp(ctl, 4);
    const bool lin_fail = ctl->lin_fail != 0;
    const double* __restrict__ H = H0 + small_sel(ctl, alt_off);
    const double* __restrict__ g = g0 + small_sel(ctl, alt_off);
    if (FUSE && (int)blockIdx.x >= nb_e) {
        const int f = ((int)blockIdx.x - nb_e) * blockDim.x + threadIdx.x;
        if (f >= n_f)
            return;
        const int p = f_off_pose + f;
        double d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            d[k] = lin_fail ? 0.0 : -yf[(row_of ? row_of[f] : 6 * f) + k] * scale[6 * (int64_t)p + k];
        candidate_for_pose(pv, p, d, delta, H, g, active, pose_part);
        return;
    }
    // One workgroup per eliminated pose: its four waves split the columns of the pose's six rows of Z (one wave per
    // pose left half of the SIMDs idle and a ten-deep chain of dependent load rounds: 10.3 us), wave 0 finishes.
    const int e = (int)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (e >= n_e)
        return;
    const bool owned = pose_task[e + 1] > pose_task[e];
    if (!owned || lin_fail) {
        if (wave != 0)
            return;
        if (lane < 6)
            step_comm[6 * (int64_t)e + lane] = 0.0;
        if (lane == 0) {
            part_cross[e] = 0.0;
            if (FUSE) {
                const double d[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
                candidate_for_pose(pv, e_off_pose + e, d, delta, H, g, active, pose_part);
            }
        }
        return;
    }
    __shared__ double s_acc[4][6];
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    if (SPARSE) {
        const int es = e_start[e], rs = 6 * (e_start[e + 1] - es);
        const double* zp = Z + 36 * (int64_t)es;   // the pose's blocks: [observation][6][6]
        for (int t = (int)threadIdx.x; t < rs; t += 256) {
            const int k = t / 6, c = t - 6 * k;
            const int fo = e_other[es + k];
            const double yv = yf[(row_of ? row_of[fo] : 6 * fo) + c];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                acc[i] += zp[36 * k + 6 * c + i] * yv;   // Z(i, c) of block k, column-major
        }
    } else {
        const double* zr = Z + (int64_t)(6 * e) * ldz;
        // 16-byte loads: n_red = 6 n_f is even and every row of Z starts 16-byte aligned (ldz is even)
        for (int c = 2 * (int)threadIdx.x; c < n_red; c += 512) {
            const double2 yv = *reinterpret_cast<const double2*>(yf + c);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double2 zv = *reinterpret_cast<const double2*>(zr + (int64_t)i * ldz + c);
                acc[i] += zv.x * yv.x + zv.y * yv.y;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double a = wave_sum(acc[i]);
        if (lane == 0)
            s_acc[wave][i] = a;
    }
    __syncthreads();
    if (wave != 0)
        return;
    double v[6];
    double cr = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double a = (s_acc[0][i] + s_acc[1][i]) + (s_acc[2][i] + s_acc[3][i]);
        v[i] = ze[6 * (int64_t)e + i] - a;
        cr += v[i] * a;
    }
    const double* L = Le + 36 * (int64_t)e;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = v[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k)
            s -= L[6 * k + i] * v[k];
        v[i] = s / L[6 * i + i];
    }
    double d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
        d[i] = -v[i] * scale[6 * (int64_t)(e_off_pose + e) + i];
    if (lane < 6) {
        double out = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            out = (lane == i) ? d[i] : out;
        step_comm[6 * (int64_t)e + lane] = out;
    }
    if (lane == 0) {
        part_cross[e] = cr;
        if (FUSE)
            candidate_for_pose(pv, e_off_pose + e, d, delta, H, g, active, pose_part);
    }
}

// World > 1 (the eliminated family's step is all-reduced first): the candidate of every pose in a launch of its own.
__global__ void k_candidate(LmCtl* ctl, PoseViews pv, int n_e, int e_off_pose, int f_off_pose,
                            const double* __restrict__ step_comm, const double* __restrict__ yf,
                            const double* __restrict__ scale, double* __restrict__ delta,
                            const double* __restrict__ H, const double* __restrict__ g,
                            const int32_t* __restrict__ active, double* __restrict__ pose_part,
                            const int32_t* __restrict__ row_of)
{
    // the summed votes of k_backsub: some rank's factorisation gave up waiting in this pass -> every rank pauses here
    // (the ranks must take the same decisions and make the same collective calls; the hosts redo the pass together)
    const bool remote = step_comm[7 * (int64_t)n_e] > 0.0;
    if (remote && blockIdx.x == 0 && threadIdx.x == 0 && ctl->done != 1) {
        if (ctl->done == 0)
            atomicOr(&ctl->sync_timeout, 4);
        atomicExch(&ctl->done, 2);
    }
    if (ctl->done || remote)
        return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_pose = pv.n_cams + pv.n_tags;
    if (p >= n_pose)
        return;
    const bool is_e = (p >= e_off_pose) && (p < e_off_pose + n_e);
    double d[6];
    for (int k = 0; k < 6; ++k) {
        if (ctl->lin_fail)
            d[k] = 0.0;
        else if (is_e)
            d[k] = step_comm[6 * (int64_t)(p - e_off_pose) + k];
        else
            d[k] = -yf[(row_of ? row_of[p - f_off_pose] : 6 * (p - f_off_pose)) + k] * scale[6 * (int64_t)p + k];
    }
    candidate_for_pose(pv, p, d, delta, H, g, active, pose_part);
}

// Start of an LM loop in ONE launch: the control block (passed by value), the poses vmm_ba_set_state staged in pinned
// host memory (read over the bus: 39 KB at 500 x 200) into the state, and the state into the candidate buffers -- five
// stream-ordered copies of a few microseconds each before.
__global__ __launch_bounds__(256) void k_begin_loop(LmCtl* ctl, const LmCtl init, const int n_cam7, const int n_tag7,
                                                    const double* __restrict__ stage_cam,
                                                    const double* __restrict__ stage_tag, double* __restrict__ cam_qt,
                                                    double* __restrict__ tag_qt, double* __restrict__ cam_cand,
                                                    double* __restrict__ tag_cand)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        *ctl = init;
    if (i < n_cam7) {
        double v;
        if (stage_cam) {
            v = stage_cam[i];
            cam_qt[i] = v;
        } else {
            v = cam_qt[i];
        }
        cam_cand[i] = v;
    } else if (i < n_cam7 + n_tag7) {
        const int k = i - n_cam7;
        double v;
        if (stage_tag) {
            v = stage_tag[k];
            tag_qt[k] = v;
        } else {
            v = tag_qt[k];
        }
        tag_cand[k] = v;
    }
}

void launch_begin_loop(Engine& e, const LmCtl& init)
{
    const int n_cam7 = 7 * e.n_cams, n_tag7 = 7 * e.n_tags;
    const double* sc = e.dirty_cam ? e.pose_stage_dev : nullptr;
    const double* st = e.dirty_tag ? e.pose_stage_dev + n_cam7 : nullptr;
    hipLaunchKernelGGL(k_begin_loop, dim3((n_cam7 + n_tag7 + 255) / 256), dim3(256), 0, e.stream, e.ctl, init, n_cam7,
                       n_tag7, sc, st, e.cam_qt, e.tag_qt, e.cam_cand, e.tag_cand);
}

// Diagnostic (vmm_ba_pose_plus): the Plus the LM loop applies, on caller-supplied poses and tangent steps.
__global__ void k_pose_plus(int64_t n, const double* __restrict__ qt, const double* __restrict__ delta,
                            double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double x[7], d[6], o[7];
    for (int k = 0; k < 7; ++k)
        x[k] = qt[7 * i + k];
    for (int k = 0; k < 6; ++k)
        d[k] = delta[6 * i + k];
    pose_plus(x, d, o);
    for (int k = 0; k < 7; ++k)
        out[7 * i + k] = o[k];
}

void launch_pose_plus(hipStream_t st, int64_t n, const double* qt, const double* delta, double* out)
{
    hipLaunchKernelGGL(k_pose_plus, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, qt, delta, out);
}

// ---- launchers -----------------------------------------------------------------------------------

static PoseViews views(Engine& e)
{
    PoseViews pv;
    pv.n_cams = e.n_cams;
    pv.n_tags = e.n_tags;
    pv.tag_euclid = e.points ? 1 : 0;
    pv.cam_qt = e.cam_qt;
    pv.tag_qt = e.tag_qt;
    pv.cam_cand = e.cam_cand;
    pv.tag_cand = e.tag_cand;
    return pv;
}

void launch_control(Engine& e)
{
    const bool single = !e.multi;
    DecideArgs dz;
    dz.pose_part = e.pose_part;
    // world > 1: k_backsub leaves its per-pose cross terms behind the steps in the all-reduced buffer
    dz.cross_parts = single ? e.part_cross : e.step_comm + 6 * (size_t)e.n_e;
    dz.n_cross = e.n_e;
    // candidate cost: the per-pose costs of the evaluation at the candidate (world > 1: all-reduced behind the staging
    // copy), summed in pose order
    dz.cost_parts = single ? e.part_cost : e.ev_pose_cost;
    dz.n_cost = e.n_e;
    dz.pose_gm = single ? e.pose_gm : (const double*)nullptr;
    // one GPU: nothing to copy (src == dst), the accepted evaluation's blocks are selected through w_which;
    // world > 1: the all-reduced staging copy becomes the working copy
    hipLaunchKernelGGL(k_control, dim3(1), dim3(1024), 0, e.stream, e.ctl, views(e), dz,
                       single ? (const double*)e.small : (const double*)e.small_stage, e.small, e.small_count, e.H_cam,
                       e.g_cam, e.small_alt_off, e.scale, e.active, e.diag, e.D2, e.trace);
}

// One GPU: back-substitution and the candidates of all poses in one launch (launch_candidate is then a no-op).
void launch_backsub(Engine& e)
{
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const int nb_e = e.n_e;   // one workgroup per eliminated pose
    const double* const Zp = e.sparse_schur ? e.Zc : e.Z;
    // world > 1: the per-pose cross terms travel with the steps (one all-reduce, summed over the poses by k_control)
    double* const crossp = e.multi ? e.step_comm + 6 * (size_t)e.n_e : e.part_cross;
#define VMM_BACKSUB(FUSE, SP, GRID)                                                                                      \
    hipLaunchKernelGGL((k_backsub<FUSE, SP>), dim3(GRID), dim3(256), 0, e.stream, e.ctl, e.n_e, e_off, e.ordE.pose_task,   \
                       Zp, e.ldz, e.n_red, e.yf, e.Le, e.ze, e.scale, e.step_comm, crossp, views(e), f_off, e.n_f,        \
                       nb_e, e.delta, e.H_cam, e.g_cam, e.small_alt_off, e.active, e.pose_part,                           \
                       (const int32_t*)e.ordE.start, (const int32_t*)e.ordE.other,                                         \
                       (const int32_t*)((e.sparse_schur && e.explicit_pairs) ? e.row_of : nullptr))
    if (e.multi) {
        if (e.sparse_schur)
            VMM_BACKSUB(false, true, nb_e);
        else
            VMM_BACKSUB(false, false, nb_e);
        return;
    }
    const int nb_f = (e.n_f + 255) / 256;
    if (e.sparse_schur)
        VMM_BACKSUB(true, true, nb_e + nb_f);
    else
        VMM_BACKSUB(true, false, nb_e + nb_f);
#undef VMM_BACKSUB
}

void launch_candidate(Engine& e)
{
    if (!e.multi)
        return;   // formed by launch_backsub
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const int n_pose = e.n_cams + e.n_tags;
    hipLaunchKernelGGL(k_candidate, dim3((n_pose + 63) / 64), dim3(64), 0, e.stream, e.ctl, views(e), e.n_e, e_off,
                       f_off, e.step_comm, e.yf, e.scale, e.delta, e.H_cam, e.g_cam, e.active, e.pose_part,
                       (const int32_t*)((e.sparse_schur && e.explicit_pairs) ? e.row_of : nullptr));
}

// Touches every kernel of this file once (vmm_ba_create): the code object is loaded and the kernel's resources
// are known before any launch is recorded into a hipGraph (nothing may be loaded lazily under stream capture).
int preload_lm_kernels()
{
    hipFuncAttributes at;
    int bad = 0;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_control)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_begin_loop)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsub<true, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsub<false, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsub<true, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_backsub<false, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_candidate)) != hipSuccess;
    return bad;
}

} // namespace vmm
