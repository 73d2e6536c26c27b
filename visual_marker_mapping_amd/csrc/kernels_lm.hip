// Device-resident trust-region control (gfx950): the policy of Ceres' TrustRegionMinimizer +
// LevenbergMarquardtStrategy that the reference runs with default options through ceres::Solve
// (src/TagReconstructor.cpp:725-738), restated per SURVEY.md Appendix A.4.  All scalars of the loop
// live in one LmCtl block in HBM; every kernel of an iteration starts by reading its `done` flag, so
// the host can enqueue iterations ahead and only polls.
#include <float.h>

#include "engine.hpp"

namespace vmm {

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int m = blockDim.x / 2; m >= 1; m >>= 1) {
        if (tid < m)
            sh[tid] += sh[tid + m];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_max(double v, double* sh)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int m = blockDim.x / 2; m >= 1; m >>= 1) {
        if (tid < m)
            sh[tid] = fmax(sh[tid], sh[tid + m]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

struct PoseViews {
    int n_cams, n_tags;
    double* cam_qt;
    double* tag_qt;
    double* cam_cand;
    double* tag_cand;
};

__device__ __forceinline__ double* pose_ptr(const PoseViews& v, int p, bool cand)
{
    if (p < v.n_cams)
        return (cand ? v.cam_cand : v.cam_qt) + 7 * (int64_t)p;
    return (cand ? v.tag_cand : v.tag_qt) + 7 * (int64_t)(p - v.n_cams);
}

// After an evaluation at x: cost, Jacobi scaling (first evaluation only), gradient max-norm.
// TrustRegionMinimizer::IterationZero / EvaluateGradientAndJacobian.
__global__ __launch_bounds__(256) void k_post_eval(LmCtl* ctl, PoseViews pv, const double* __restrict__ src,
                                                   double* __restrict__ dst, size_t small_count,
                                                   const double* __restrict__ H, const double* __restrict__ g,
                                                   const double* __restrict__ cost_slot,
                                                   double* __restrict__ scale, int32_t* __restrict__ active)
{
    if (ctl->done || !ctl->need_jacobian)
        return;
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int n_pose = pv.n_cams + pv.n_tags;
    // multi-GPU: the all-reduced staging buffer becomes the working copy
    if (src != dst) {
        for (size_t i = tid; i < small_count; i += 256)
            dst[i] = src[i];
        __syncthreads();
    }
    const bool first = ctl->first_eval != 0;
    const bool jacobi = ctl->jacobi_scaling != 0;
    double xn = 0.0, gm = 0.0;
    for (int p = tid; p < n_pose; p += 256) {
        const double* Hp = H + 36 * (int64_t)p;
        if (first) {
            // blocks with zero Jacobian columns (constant origin tag, poses without observations)
            // are not part of Ceres' reduced program
            active[p] = (Hp[0] + Hp[7] + Hp[14]) > 0.0 ? 1 : 0;
            for (int k = 0; k < 6; ++k)
                scale[6 * (int64_t)p + k] = jacobi ? 1.0 / (1.0 + sqrt(Hp[7 * k])) : 1.0;
        }
        if (active[p]) {
            const double* x = pose_ptr(pv, p, false);
            if (first)
                for (int k = 0; k < 7; ++k)
                    xn += x[k] * x[k];
            double ng[6], xp[7];
            for (int k = 0; k < 6; ++k)
                ng[k] = -g[6 * (int64_t)p + k];
            pose_plus(x, ng, xp);
            for (int k = 0; k < 7; ++k)
                gm = fmax(gm, fabs(x[k] - xp[k]));
        }
    }
    xn = block_sum(xn, sh);
    gm = block_max(gm, sh);
    if (tid == 0) {
        const double cost = cost_slot[0];
        ctl->x_cost = cost;
        ctl->need_jacobian = 0;
        ctl->num_jac_evals++;
        if (first) {
            ctl->first_eval = 0;
            ctl->initial_cost = cost;
            ctl->x_norm = sqrt(xn);
            ctl->cur.iteration = 0;
            ctl->cur.step_is_valid = 1;
            ctl->cur.step_is_successful = 1;
        }
        ctl->cur.cost = cost;
        ctl->cur.gradient_max_norm = gm;
        if (!isfinite(cost)) {
            ctl->done = 1;
            ctl->termination = VMM_BA_FAILURE;
        }
    }
}

// FinalizeIterationAndCheckIfMinimizerCanContinue, then the first half of
// LevenbergMarquardtStrategy::ComputeStep (the LM diagonal).
__global__ __launch_bounds__(256) void k_lm_begin(LmCtl* ctl, int n_tan, const double* __restrict__ H,
                                                  const double* __restrict__ scale, double* __restrict__ diag,
                                                  double* __restrict__ D2, vmm_ba_iteration* __restrict__ trace)
{
    if (ctl->done)
        return;
    __shared__ int s_stop, s_reuse;
    __shared__ double s_radius;
    if (threadIdx.x == 0) {
        vmm_ba_iteration cur = ctl->cur;
        cur.trust_region_radius = ctl->radius;
        if (cur.step_is_successful)
            ctl->num_successful++;
        else
            ctl->num_unsuccessful++;
        if (ctl->records < ctl->trace_capacity)
            trace[ctl->records] = cur;
        ctl->records++;
        int stop = 0;
        if (cur.iteration >= ctl->max_num_iterations) {
            stop = 1;
            ctl->termination = VMM_BA_NO_CONVERGENCE;
        } else if (cur.step_is_successful && cur.gradient_max_norm <= ctl->gradient_tolerance) {
            stop = 1;
            ctl->termination = VMM_BA_CONVERGENCE;
        } else if (ctl->radius <= ctl->min_radius) {
            stop = 1;
            ctl->termination = VMM_BA_CONVERGENCE;
        }
        if (stop) {
            ctl->done = 1;
        } else {
            const double gmax = cur.gradient_max_norm;
            vmm_ba_iteration z;
            z.iteration = cur.iteration + 1;
            z.step_is_valid = 0;
            z.step_is_successful = 0;
            z.reserved = 0;
            z.cost = ctl->x_cost;
            z.cost_change = 0.0;
            z.gradient_max_norm = gmax;  // carried until the next successful step
            z.step_norm = 0.0;
            z.relative_decrease = 0.0;
            z.trust_region_radius = 0.0;
            z.model_cost_change = 0.0;
            ctl->cur = z;
            ctl->iteration = z.iteration;
            ctl->lin_fail = 0;
            ctl->num_lm_iterations++;
        }
        s_stop = stop;
        s_reuse = ctl->reuse_diagonal;
        s_radius = ctl->radius;
        if (!stop)
            ctl->reuse_diagonal = 1;
    }
    __syncthreads();
    if (s_stop)
        return;
    const double lo = ctl->min_lm_diagonal, hi = ctl->max_lm_diagonal;
    for (int c = threadIdx.x; c < n_tan; c += 256) {
        double d;
        if (!s_reuse) {
            const int p = c / 6, k = c % 6;
            const double s = scale[c];
            d = s * s * H[36 * (int64_t)p + 7 * k];   // squared column norm of the scaled Jacobian
            d = fmin(fmax(d, lo), hi);
            diag[c] = d;
        } else {
            d = diag[c];
        }
        const double lm = sqrt(d / s_radius);         // lm_diagonal_ = sqrt(diagonal_ / radius_)
        D2[c] = lm * lm;
    }
}

// Back-substitution of the eliminated family: y_e = L_e^{-T} (z_e - Z_e y_f), delta_e = -s_e y_e.
// One wave per eliminated pose.
__global__ __launch_bounds__(256) void k_backsub(const LmCtl* ctl, int n_e, int e_off_pose,
                                                 const int32_t* __restrict__ pose_task,
                                                 const double* __restrict__ Z, int ldz, int n_red,
                                                 const double* __restrict__ yf, const double* __restrict__ Le,
                                                 const double* __restrict__ ze, const double* __restrict__ scale,
                                                 double* __restrict__ step_comm)
{
    if (ctl->done)
        return;
    const int e = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= n_e)
        return;
    const bool owned = pose_task[e + 1] > pose_task[e];
    if (!owned || ctl->lin_fail) {
        if (lane < 6)
            step_comm[6 * (int64_t)e + lane] = 0.0;
        return;
    }
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    const double* zr = Z + (int64_t)(6 * e) * ldz;
    for (int c = lane; c < n_red; c += 64) {
        const double yv = yf[c];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[i] += zr[(int64_t)i * ldz + c] * yv;
    }
    double v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
        v[i] = ze[6 * (int64_t)e + i] - wave_sum(acc[i]);
    const double* L = Le + 36 * (int64_t)e;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = v[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k)
            s -= L[6 * k + i] * v[k];
        v[i] = s / L[6 * i + i];
    }
    if (lane < 6) {
        double out = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            out = (lane == i) ? v[i] : out;
        step_comm[6 * (int64_t)e + lane] = -out * scale[6 * (int64_t)(e_off_pose + e) + lane];
    }
}

// Cross term of the model cost: per observation delta_e^T W_ef delta_f (E order), wave partials.
__global__ __launch_bounds__(256) void k_cross(const LmCtl* ctl, const Task* __restrict__ tasks, int n_tasks,
                                               const int32_t* __restrict__ other, const double* __restrict__ W,
                                               int64_t n_pad, const double* __restrict__ step_comm,
                                               const double* __restrict__ yf, const double* __restrict__ scale,
                                               int f_off_pose, double* __restrict__ part)
{
    if (ctl->done)
        return;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n_tasks)
        return;
    const Task t = tasks[wave];
    const int64_t i = (int64_t)t.begin + lane;
    double v = 0.0;
    if (i < t.end && !ctl->lin_fail) {
        const int f = other[i];
        double de[6], df[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            de[k] = step_comm[6 * (int64_t)t.pose + k];
            df[k] = -yf[6 * f + k] * scale[6 * (int64_t)(f_off_pose + f) + k];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double r = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b)
                r += W[(int64_t)(6 * a + b) * n_pad + i] * df[b];
            v += de[a] * r;
        }
    }
    v = wave_sum(v);
    if (lane == 0)
        part[wave] = v;
}

// delta (unscaled tangent step) for every pose and the candidate x+ = Plus(x, delta).
// TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost (first half).
__global__ void k_candidate(const LmCtl* ctl, PoseViews pv, int n_e, int e_off_pose, int f_off_pose,
                            const double* __restrict__ step_comm, const double* __restrict__ yf,
                            const double* __restrict__ scale, double* __restrict__ delta)
{
    if (ctl->done)
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
            d[k] = -yf[6 * (int64_t)(p - f_off_pose) + k] * scale[6 * (int64_t)p + k];
        delta[6 * (int64_t)p + k] = d[k];
    }
    double out[7];
    pose_plus(pose_ptr(pv, p, false), d, out);
    double* c = pose_ptr(pv, p, true);
    for (int k = 0; k < 7; ++k)
        c[k] = out[k];
}

// Step validation, convergence tests, acceptance and radius update:
// ComputeTrustRegionStep (model cost), ParameterToleranceReached, FunctionToleranceReached,
// IsStepSuccessful, HandleSuccessfulStep / HandleUnsuccessfulStep / HandleInvalidStep,
// LevenbergMarquardtStrategy::StepAccepted / StepRejected.
__global__ __launch_bounds__(256) void k_decide(LmCtl* ctl, PoseViews pv, const double* __restrict__ H,
                                                const double* __restrict__ g, const double* __restrict__ delta,
                                                const int32_t* __restrict__ active,
                                                const double* __restrict__ cross_slot,
                                                const double* __restrict__ cand_cost_slot)
{
    if (ctl->done)
        return;
    __shared__ double sh[256];
    __shared__ int s_accept;
    const int tid = threadIdx.x;
    const int n_pose = pv.n_cams + pv.n_tags;
    double gd = 0.0, quad = 0.0, sn = 0.0, xn = 0.0, bad = 0.0;
    for (int p = tid; p < n_pose; p += 256) {
        const double* d = delta + 6 * (int64_t)p;
        const double* Hp = H + 36 * (int64_t)p;
        for (int a = 0; a < 6; ++a) {
            if (!isfinite(d[a]))
                bad = 1.0;
            gd += d[a] * g[6 * (int64_t)p + a];
            double r = 0.0;
            for (int b = 0; b < 6; ++b)
                r += Hp[6 * a + b] * d[b];
            quad += d[a] * r;
        }
        if (active[p]) {
            const double* x = pose_ptr(pv, p, false);
            const double* c = pose_ptr(pv, p, true);
            for (int k = 0; k < 7; ++k) {
                const double df = x[k] - c[k];
                sn += df * df;
                xn += c[k] * c[k];
            }
        }
    }
    gd = block_sum(gd, sh);
    quad = block_sum(quad, sh);
    sn = block_sum(sn, sh);
    xn = block_sum(xn, sh);
    bad = block_max(bad, sh);
    if (tid == 0) {
        int accept = 0;
        const bool lin_fail = ctl->lin_fail != 0 || bad != 0.0;
        // model_cost_change = -(J d)^T (r + J d / 2) = -d^T g - 1/2 d^T H d   (unscaled coordinates)
        const double mcc = lin_fail ? 0.0 : -gd - 0.5 * (quad + 2.0 * cross_slot[0]);
        ctl->model_cost_change = mcc;
        ctl->cur.model_cost_change = mcc;
        const bool valid = !lin_fail && (mcc > 0.0);
        ctl->cur.step_is_valid = valid ? 1 : 0;
        if (!valid) {
            // HandleInvalidStep
            ctl->num_invalid++;
            if (ctl->num_invalid >= ctl->max_invalid) {
                ctl->done = 1;
                ctl->termination = VMM_BA_FAILURE;
            } else {
                ctl->radius = ctl->radius / ctl->decrease_factor;
                ctl->decrease_factor *= 2.0;
                ctl->reuse_diagonal = 1;
                ctl->cur.cost = ctl->x_cost;
                ctl->cur.step_is_successful = 0;
            }
        } else {
            ctl->num_invalid = 0;
            double cand = cand_cost_slot[0];
            ctl->num_cost_evals++;
            if (!isfinite(cand))
                cand = DBL_MAX;
            ctl->cand_cost = cand;
            const double step_norm = sqrt(sn);
            ctl->cur.step_norm = step_norm;
            const double x_cost = ctl->x_cost;
            const double cost_change = x_cost - cand;
            if (step_norm <= ctl->parameter_tolerance * (ctl->x_norm + ctl->parameter_tolerance)) {
                ctl->done = 1;   // ParameterToleranceReached: return without pushing this record
                ctl->termination = VMM_BA_CONVERGENCE;
            } else if (fabs(cost_change) <= ctl->function_tolerance * x_cost) {
                ctl->cur.cost_change = cost_change;
                ctl->done = 1;   // FunctionToleranceReached
                ctl->termination = VMM_BA_CONVERGENCE;
            } else {
                ctl->cur.cost_change = cost_change;
                const double rd = (cand >= DBL_MAX) ? -DBL_MAX : cost_change / mcc;
                ctl->cur.relative_decrease = rd;
                if (rd > ctl->min_relative_decrease) {
                    accept = 1;
                    const double q = 2.0 * rd - 1.0;
                    double den = 1.0 - q * q * q;
                    den = den < 1.0 / 3.0 ? 1.0 / 3.0 : den;
                    double r = ctl->radius / den;
                    ctl->radius = r > ctl->max_radius ? ctl->max_radius : r;
                    ctl->decrease_factor = 2.0;
                    ctl->reuse_diagonal = 0;
                    ctl->need_jacobian = 1;
                    ctl->x_norm = sqrt(xn);
                    ctl->cur.step_is_successful = 1;
                } else {
                    ctl->cur.step_is_successful = 0;
                    ctl->cur.cost = cand;
                    ctl->radius = ctl->radius / ctl->decrease_factor;
                    ctl->decrease_factor *= 2.0;
                    ctl->reuse_diagonal = 1;
                }
            }
        }
        s_accept = accept;
    }
    __syncthreads();
    if (s_accept) {
        for (int i = tid; i < 7 * pv.n_cams; i += 256)
            pv.cam_qt[i] = pv.cam_cand[i];
        for (int i = tid; i < 7 * pv.n_tags; i += 256)
            pv.tag_qt[i] = pv.tag_cand[i];
    }
}

// Zeroes a staging buffer when this iteration did not evaluate (multi-GPU only), so the
// unconditional all-reduce that follows is a no-op for the consumer.
__global__ void k_zero_unless_eval(const LmCtl* ctl, double* buf, size_t n)
{
    if (ctl->done || ctl->need_jacobian)
        return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        buf[i] = 0.0;
}

// ---- launchers -----------------------------------------------------------------------------------

static PoseViews views(Engine& e)
{
    PoseViews pv;
    pv.n_cams = e.n_cams;
    pv.n_tags = e.n_tags;
    pv.cam_qt = e.cam_qt;
    pv.tag_qt = e.tag_qt;
    pv.cam_cand = e.cam_cand;
    pv.tag_cand = e.tag_cand;
    return pv;
}

void launch_zero_unless_eval(Engine& e, double* buf, size_t n)
{
    hipLaunchKernelGGL(k_zero_unless_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e.stream, e.ctl, buf, n);
}

void launch_post_eval(Engine& e, const double* src)
{
    hipLaunchKernelGGL(k_post_eval, dim3(1), dim3(256), 0, e.stream, e.ctl, views(e), src, e.small, e.small_count,
                       e.H_cam, e.g_cam, e.cost_slot, e.scale, e.active);
}

void launch_lm_begin(Engine& e)
{
    hipLaunchKernelGGL(k_lm_begin, dim3(1), dim3(256), 0, e.stream, e.ctl, 6 * (e.n_cams + e.n_tags), e.H_cam, e.scale,
                       e.diag, e.D2, e.trace);
}

void launch_backsub(Engine& e)
{
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    hipLaunchKernelGGL(k_backsub, dim3((e.n_e + 3) / 4), dim3(256), 0, e.stream, e.ctl, e.n_e, e_off, e.ordE.pose_task,
                       e.Z, e.ldz, e.n_red, e.yf, e.Le, e.ze, e.scale, e.step_comm);
    if (e.ordE.n_tasks > 0)
        hipLaunchKernelGGL(k_cross, dim3((e.ordE.n_tasks + 3) / 4), dim3(256), 0, e.stream, e.ctl, e.ordE.tasks,
                           e.ordE.n_tasks, e.ordE.other, e.W, e.ordE.n_pad, e.step_comm, e.yf, e.scale, f_off,
                           e.part_cross);
}

void launch_candidate(Engine& e)
{
    const int e_off = e.elim_cams ? 0 : e.n_cams;
    const int f_off = e.elim_cams ? e.n_cams : 0;
    const int n_pose = e.n_cams + e.n_tags;
    hipLaunchKernelGGL(k_candidate, dim3((n_pose + 127) / 128), dim3(128), 0, e.stream, e.ctl, views(e), e.n_e, e_off,
                       f_off, e.step_comm, e.yf, e.scale, e.delta);
}

void launch_decide(Engine& e)
{
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, e.stream, e.ctl, views(e), e.H_cam, e.g_cam, e.delta, e.active,
                       e.step_comm + 6 * (size_t)e.n_e, e.cost_comm);
}

} // namespace vmm
