// Residual / Jacobian evaluation kernels (gfx950).
//
// Replaces, for all 4*N_obs corner blocks at once, what Ceres' Evaluator does for the reference:
// the functor TagReconstructionCostFunction::operator() (include/visual_marker_mapping/
// TagReconstructionCostFunction.h:101-159) under AutoDiffCostFunction<...,2,3,4,3,4> (:167) with
// QuaternionParameterization and HuberLoss(1.0) (src/TagReconstructor.cpp:661,721), followed by the
// J^T J / J^T r formation of the sparse normal-Cholesky solver (src/TagReconstructor.cpp:737-738).
//
// Work decomposition: observations are stored twice, sorted by camera and sorted by tag.  One wave
// = one Task = up to 64 consecutive observations of ONE pose, lane = observation (coalesced SoA
// loads of the 8 pixel coordinates).  The pass over a family accumulates that family's diagonal
// 6x6 block and gradient in registers and reduces them with a fixed butterfly, so the sums are
// deterministic and need no atomics; the pass over the eliminated family additionally writes the
// 6x6 off-diagonal block W = J_e^T J_f of every observation.
#include "engine.hpp"

namespace vmm {

struct EvalArgs {
    Intrinsics K;
    const Task* tasks;
    int n_tasks;
    const int32_t* other;
    const double* px;
    int64_t n_pad;
    const double* own_pose;    // poses of the sorted family
    const double* other_pose;  // poses of the other family
    const double* tag_wh;
    int fixed_tag;
    int fixed_shift;           // point landmarks: two 6-dof blocks (point pairs) per tag, block >> 1 == fixed_tag
    int robustify;
    double huber_a;
    double* part;              // [n_tasks][kPart]
    void* W;                   // [36][n_pad] of AT (double or float) or null
    void* W_alt;               // LM loop: the evaluation at the candidate writes the buffer LmCtl::w_which does NOT name
    const int32_t* caller;     // position of each sorted observation in the caller's order
    const uint8_t* mask;       // [n_obs] caller order: 0 = observation switched off (vmm_ba_set_observation_mask)
    const LmCtl* ctl;          // LM loop: returns at once when the loop is over; null: always run
};

__device__ __forceinline__ int tri(int a, int b) { return a * (a + 1) / 2 + b; }

// AT: accumulation/storage type of the J^T J blocks (double, or float for VMM_BA_PRECISION_F32_ACCUM);
// residuals, the cost and the gradient J^T r are always f64.
// POINTS: the landmark family holds pairs of free 3-D points (vmm_ba_create_options.landmarks ==
// VMM_BA_LANDMARK_POINTS): an observation is the two corner observations of one pair, each corner is
// OpenCVReprojectionError (CostFunction.h:21-68) and touches three of the block's six columns.
template <bool OWN_IS_CAM, bool WRITE_W, typename AT, bool POINTS = false>
__device__ __forceinline__ void eval_body(const EvalArgs& a, const int wave)
{
    const int lane = threadIdx.x & 63;
    if (wave >= a.n_tasks)
        return;
    const Task t = a.tasks[wave];
    const int64_t i = (int64_t)t.begin + lane;
    const bool valid = i < t.end;
    const int64_t is = valid ? i : t.begin;
    const int o = a.other[is];
    // LM loop: W of the candidate goes to the buffer that does not hold W at x
    AT* __restrict__ const Wout = static_cast<AT*>((WRITE_W && a.W_alt && a.ctl && a.ctl->w_which == 0) ? a.W_alt : a.W);

    const int tag_idx = OWN_IS_CAM ? o : t.pose;
    const int cam_idx = OWN_IS_CAM ? t.pose : o;
    const double* camq = (OWN_IS_CAM ? a.own_pose : a.other_pose) + 7 * (int64_t)cam_idx;
    const double* tagq = (OWN_IS_CAM ? a.other_pose : a.own_pose) + 7 * (int64_t)tag_idx;
    Rigid cam, tag;
    double pt[6];
    if (POINTS) {
        load_rigid<false>(camq, cam);   // UnitQuaternionRotatePoint: the quaternion as it is
#pragma unroll
        for (int k = 0; k < 6; ++k)
            pt[k] = tagq[k];
    } else {
        load_rigid<true>(camq, cam);
        load_rigid<true>(tagq, tag);
    }
    const double hw = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx], hh = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx + 1];
    // a constant (origin) tag contributes no Jacobian columns (src/TagReconstructor.cpp:669-673, :494-497)
    const double tag_on = ((tag_idx >> a.fixed_shift) == a.fixed_tag) ? 0.0 : 1.0;
    // switched-off observations are selected out, never multiplied out (their poses are parked defaults)
    const bool on = valid && a.mask[a.caller[is]];

    AT H[21];
    double g[6], cost = 0.0;
    AT Wacc[WRITE_W ? 36 : 1];
#pragma unroll
    for (int k = 0; k < 21; ++k)
        H[k] = (AT)0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        g[k] = 0.0;
    if (WRITE_W) {
#pragma unroll
        for (int k = 0; k < 36; ++k)
            Wacc[k] = (AT)0;
    }

    constexpr bool NEED_JC = OWN_IS_CAM || WRITE_W;
    constexpr bool NEED_JT = !OWN_IS_CAM || WRITE_W;
    // not unrolled: four copies of the corner's temporaries cost 30 more spilled VGPRs (scratch 140 -> 20 B per lane
    // at two waves per SIMD) and the scratch traffic that goes with them; 31.2 -> 30.0 us for the evaluation
#pragma unroll 1
    for (int c = 0; c < (POINTS ? 2 : 4); ++c) {
        // LL, LR, UR, UL (include/visual_marker_mapping/TagReconstructor.h:47-50)
        const double sx = (c == 1 || c == 2) ? hw : -hw;
        const double sy = (c >= 2) ? hh : -hh;
        const double u = a.px[(2 * c) * a.n_pad + is];
        const double v = a.px[(2 * c + 1) * a.n_pad + is];
        CornerEval e;
        if (POINTS) {
            eval_point<NEED_JC, NEED_JT>(a.K, cam, camq, pt[3 * c], pt[3 * c + 1], pt[3 * c + 2], u, v, e, on);
            if (NEED_JT) {
                // the corner's point owns columns 3c .. 3c+2 of the pair's block
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double j0 = e.jt[r][0], j1 = e.jt[r][1], j2 = e.jt[r][2];
                    e.jt[r][0] = c == 0 ? j0 : 0.0;
                    e.jt[r][1] = c == 0 ? j1 : 0.0;
                    e.jt[r][2] = c == 0 ? j2 : 0.0;
                    e.jt[r][3] = c == 0 ? 0.0 : j0;
                    e.jt[r][4] = c == 0 ? 0.0 : j1;
                    e.jt[r][5] = c == 0 ? 0.0 : j2;
                }
            }
        } else {
            eval_corner<NEED_JC, NEED_JT>(a.K, cam, tag, sx, sy, u, v, e, on);
        }
        const double s = e.ru * e.ru + e.rv * e.rv;
        double rho0, wgt;
        huber(a.robustify != 0, a.huber_a, s, rho0, wgt);
        wgt = on ? wgt : 0.0;
        cost += on ? 0.5 * rho0 : 0.0;
        const double w_own = OWN_IS_CAM ? wgt : wgt * tag_on;
        const double w_oth = OWN_IS_CAM ? wgt * tag_on : wgt;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double res = (r == 0 ? e.ru : e.rv) * wgt;  // corrected residual
            double jo[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                jo[k] = (OWN_IS_CAM ? e.jc[r][k] : e.jt[r][k]) * w_own;
            AT ja[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                ja[k] = (AT)jo[k];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                g[p] += jo[p] * res;
#pragma unroll
                for (int q = 0; q <= p; ++q)
                    H[tri(p, q)] += ja[p] * ja[q];
            }
            if (WRITE_W) {
                AT jx[6];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    jx[k] = (AT)((OWN_IS_CAM ? e.jt[r][k] : e.jc[r][k]) * w_oth);
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        Wacc[6 * p + q] += ja[p] * jx[q];
            }
        }
    }
    if (WRITE_W && valid) {
#pragma unroll
        for (int k = 0; k < 36; ++k)
            Wout[(int64_t)k * a.n_pad + i] = Wacc[k];
    }
    // wave reduction of the 28 family sums (21 H + 6 g + cost) in one shared butterfly
    double red[32];
#pragma unroll
    for (int k = 0; k < 21; ++k)
        red[k] = (double)H[k];
#pragma unroll
    for (int k = 0; k < 6; ++k)
        red[21 + k] = g[k];
    red[27] = cost;
#pragma unroll
    for (int k = 28; k < 32; ++k)
        red[k] = 0.0;
    const double mine = wave_sum32(red, lane);
    const int slot = wave_sum32_index(lane);
    if (!(lane & 1) && slot < 28)
        a.part[(int64_t)wave * kPart + slot] = mine;
}

// Both family passes in one launch: workgroups [0, nb_e) run the pass over the eliminated family (writes W),
// the rest the pass over the kept family.  The passes are independent, so they share the chip.
template <bool E_IS_CAM, typename AT = double, bool POINTS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_eval_both(const EvalArgs aE, const EvalArgs aF, const int nb_e)
{
    if (aE.ctl) {
        if (aE.ctl->done)
            return;
        phase_stamp(aE.ctl, 0);
    }
    if ((int)blockIdx.x < nb_e)
        eval_body<E_IS_CAM, true, AT, POINTS>(aE, (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    else
        eval_body<!E_IS_CAM, false, AT, POINTS>(aF, (int)(((blockIdx.x - nb_e) * blockDim.x + threadIdx.x) >> 6));
}


// ------------------------------------------------------------------------------------------------
// Fused evaluation: every observation is evaluated ONCE (k_eval_both evaluates it twice: in the eliminated family's
// order for that family's blocks and W, and again in the kept family's order for the kept blocks -- Ceres evaluates
// TagReconstructionCostFunction, CostFunction.h:101-159, once per residual block).
//
// A wave owns 64 kept poses (lane = kept pose f, its pose expanded once per wave) and walks `group` eliminated poses
// e; the observation of pair (e, f) comes from a lookup table (E-order index or -1).  Per e: the eliminated pose is
// wave-uniform, the lane evaluates its observation, W goes out, the eliminated family's 28 sums (H 21, g 6, cost) are
// butterfly-reduced into the partial of (e, chunk) -- exactly the partials of the two-pass kernel when every e sees
// every f in order.  The kept family's sums stay in the lane's registers across the e's and leave once per wave:
// partF[group][28][n_f_pad] (SoA: lanes write consecutive doubles), summed over the groups in fixed order by
// k_reduce_pose.  No atomics, every sum in a fixed order.  Used when most pairs are observed (Engine::fused_eval).
struct FusedArgs {
    const int32_t* pair_obs;    // [n_e][n_f_pad] E-order observation index of (e, f), -1: not observed
    const int32_t* e_list;      // [n_e_act] eliminated poses that own an observation (here: on this rank)
    const int32_t* e_part0;     // [n_e] first partial slot of pose e (its n_chunks partials are consecutive)
    int n_e_act, n_f, n_f_pad, n_chunks, group, n_groups;
    double* partF;              // [n_groups][28][n_f_pad]
};

template <bool E_IS_CAM, typename AT, bool POINTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_eval_fused(const EvalArgs a, const FusedArgs fa)
{
    if (a.ctl) {
        if (a.ctl->done)
            return;
        phase_stamp(a.ctl, 0);
    }
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= fa.n_groups * fa.n_chunks)
        return;
    // chunk fastest: the four waves of a workgroup walk the same eliminated poses
    const int grp = wave / fa.n_chunks, c = wave - grp * fa.n_chunks;
    const int f = 64 * c + lane;
    const bool fvalid = f < fa.n_f;
    const int fs = fvalid ? f : fa.n_f - 1;
    AT* __restrict__ const Wout = static_cast<AT*>((a.W_alt && a.ctl && a.ctl->w_which == 0) ? a.W_alt : a.W);
    // the kept pose of this lane: a.other_pose is the kept family here (a = the eliminated family's arguments)
    const double* keptq = a.other_pose + 7 * (int64_t)fs;
    Rigid kept;
    double pt[6];
    if (POINTS && E_IS_CAM) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
            pt[k] = keptq[k];
    } else if (POINTS) {
        load_rigid<false>(keptq, kept);   // kept = cameras: UnitQuaternionRotatePoint, the quaternion as it is
    } else {
        load_rigid<true>(keptq, kept);
    }
    AT HF[21];
    double gF[6];
#pragma unroll
    for (int k = 0; k < 21; ++k)
        HF[k] = (AT)0;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        gF[k] = 0.0;
    const int e0 = grp * fa.group, e1 = min(e0 + fa.group, fa.n_e_act);
    // One wave per SIMD (the accumulators of both families need ~330 registers), so nothing else hides the memory
    // latency: the loads of an observation are issued one eliminated pose ahead of their use (pixels, mask) and its
    // lookup two poses ahead.
    constexpr int NPX = POINTS ? 4 : 8;
    int e_n = __builtin_amdgcn_readfirstlane(fa.e_list[e0]);
    int idx_n = fvalid ? fa.pair_obs[(int64_t)e_n * fa.n_f_pad + f] : -1;
    int e_nn = e_n, idx_nn = -1;
    if (e0 + 1 < e1) {
        e_nn = __builtin_amdgcn_readfirstlane(fa.e_list[e0 + 1]);
        idx_nn = fvalid ? fa.pair_obs[(int64_t)e_nn * fa.n_f_pad + f] : -1;
    }
    double px_n[NPX];
    int on_n;
    {
        const int64_t isn = idx_n >= 0 ? idx_n : 0;
#pragma unroll
        for (int k = 0; k < NPX; ++k)
            px_n[k] = a.px[(int64_t)k * a.n_pad + isn];
        on_n = idx_n >= 0 && a.mask[a.caller[isn]];
    }
    for (int ei = e0; ei < e1; ++ei) {
        const int e = e_n;
        const int idx = idx_n;
        const bool valid = idx >= 0;
        const bool on = on_n != 0;
        double pxc[NPX];
#pragma unroll
        for (int k = 0; k < NPX; ++k)
            pxc[k] = px_n[k];
        // prefetch: the next pose's observation, the lookup of the one after it
        e_n = e_nn;
        idx_n = idx_nn;
        if (ei + 1 < e1) {
            const int64_t isn = idx_n >= 0 ? idx_n : 0;
#pragma unroll
            for (int k = 0; k < NPX; ++k)
                px_n[k] = a.px[(int64_t)k * a.n_pad + isn];
            on_n = idx_n >= 0 && a.mask[a.caller[isn]];
        }
        if (ei + 2 < e1) {
            e_nn = __builtin_amdgcn_readfirstlane(fa.e_list[ei + 2]);
            idx_nn = fvalid ? fa.pair_obs[(int64_t)e_nn * fa.n_f_pad + f] : -1;
        }
        const int tag_idx = E_IS_CAM ? fs : e;
        const double* elimq = a.own_pose + 7 * (int64_t)e;   // wave-uniform
        Rigid elim;
        if (POINTS && !E_IS_CAM) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                pt[k] = elimq[k];
        } else if (POINTS) {
            load_rigid<false>(elimq, elim);
        } else {
            load_rigid<true>(elimq, elim);
        }
        const Rigid& cam = E_IS_CAM ? elim : kept;
        const Rigid& tag = E_IS_CAM ? kept : elim;
        const double* camq = E_IS_CAM ? elimq : keptq;
        const double hw = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx], hh = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx + 1];
        const double tag_on = ((tag_idx >> a.fixed_shift) == a.fixed_tag) ? 0.0 : 1.0;
        AT H[21];
        double g[6], cost = 0.0;
        AT Wacc[36];
#pragma unroll
        for (int k = 0; k < 21; ++k)
            H[k] = (AT)0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            g[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 36; ++k)
            Wacc[k] = (AT)0;
        // not unrolled (four copies of the corner's temporaries spill even at one wave per SIMD); the corner's pixels
        // are selected from the prefetched registers, never indexed (a runtime-indexed register array lives in scratch)
#pragma unroll 1
        for (int cn = 0; cn < (POINTS ? 2 : 4); ++cn) {
            const double sx = (cn == 1 || cn == 2) ? hw : -hw;
            const double sy = (cn >= 2) ? hh : -hh;
            double u = pxc[0], v = pxc[1];
#pragma unroll
            for (int k = 1; k < NPX / 2; ++k) {
                u = (cn == k) ? pxc[2 * k] : u;
                v = (cn == k) ? pxc[2 * k + 1] : v;
            }
            CornerEval ce;
            if (POINTS) {
                eval_point<true, true>(a.K, cam, camq, pt[3 * cn], pt[3 * cn + 1], pt[3 * cn + 2], u, v, ce, on);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double j0 = ce.jt[r][0], j1 = ce.jt[r][1], j2 = ce.jt[r][2];
                    ce.jt[r][0] = cn == 0 ? j0 : 0.0;
                    ce.jt[r][1] = cn == 0 ? j1 : 0.0;
                    ce.jt[r][2] = cn == 0 ? j2 : 0.0;
                    ce.jt[r][3] = cn == 0 ? 0.0 : j0;
                    ce.jt[r][4] = cn == 0 ? 0.0 : j1;
                    ce.jt[r][5] = cn == 0 ? 0.0 : j2;
                }
            } else {
                eval_corner<true, true>(a.K, cam, tag, sx, sy, u, v, ce, on);
            }
            const double s = ce.ru * ce.ru + ce.rv * ce.rv;
            double rho0, wgt;
            huber(a.robustify != 0, a.huber_a, s, rho0, wgt);
            wgt = on ? wgt : 0.0;
            cost += on ? 0.5 * rho0 : 0.0;
            const double w_e = E_IS_CAM ? wgt : wgt * tag_on;   // weight of the eliminated family's columns
            const double w_f = E_IS_CAM ? wgt * tag_on : wgt;   // ... of the kept family's
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const double res = (r == 0 ? ce.ru : ce.rv) * wgt;   // corrected residual
                double je[6], jf[6];
                AT ae[6], af[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    je[k] = (E_IS_CAM ? ce.jc[r][k] : ce.jt[r][k]) * w_e;
                    jf[k] = (E_IS_CAM ? ce.jt[r][k] : ce.jc[r][k]) * w_f;
                    ae[k] = (AT)je[k];
                    af[k] = (AT)jf[k];
                }
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    g[p] += je[p] * res;
                    gF[p] += jf[p] * res;
#pragma unroll
                    for (int q = 0; q <= p; ++q) {
                        H[tri(p, q)] += ae[p] * ae[q];
                        HF[tri(p, q)] += af[p] * af[q];
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        Wacc[6 * p + q] += ae[p] * af[q];
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < 36; ++k)
                Wout[(int64_t)k * a.n_pad + idx] = Wacc[k];
        }
        double red[32];
#pragma unroll
        for (int k = 0; k < 21; ++k)
            red[k] = (double)H[k];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            red[21 + k] = g[k];
        red[27] = cost;
#pragma unroll
        for (int k = 28; k < 32; ++k)
            red[k] = 0.0;
        const double mine = wave_sum32(red, lane);
        const int slot = wave_sum32_index(lane);
        if (!(lane & 1) && slot < 28)
            a.part[(int64_t)(fa.e_part0[e] + c) * kPart + slot] = mine;
    }
    if (fvalid) {
        double* pf = fa.partF + (int64_t)grp * 28 * fa.n_f_pad + f;
#pragma unroll
        for (int k = 0; k < 21; ++k)
            pf[(int64_t)k * fa.n_f_pad] = (double)HF[k];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            pf[(int64_t)(21 + k) * fa.n_f_pad] = gF[k];
    }
}

// Sums the task partials of every pose in task order and expands the packed lower triangle.
struct ReduceArgs {
    int n_pose;
    const int32_t* pose_task;   // task range per pose; null: every pose has the tasks 0 .. n_fixed-1 (fused kept family)
    int n_fixed;
    int64_t st, sk, sp;         // partial (t, k) of pose p at part[t * st + k * sk + p * sp]
    const double* part;
    double* Hout;
    double* gout;
    double* pose_cost;   // or null
    // LM loop on one GPU: the pose's share of the gradient max-norm |Plus(x, -g) - x|_inf at the evaluated poses
    // (TrustRegionMinimizer::EvaluateGradientAndJacobian), so that the single-block control kernel only takes a max
    const double* pose7; // the evaluated poses of this family (7 doubles each), or null
    int euclid;          // point landmarks: Plus is plain addition of six numbers
    double* gm_out;      // [n_pose]
};

// alt_off != 0 (LM loop on one GPU): the blocks go to the copy of the small buffer that does NOT belong to x.
__global__ void k_reduce_pose(const LmCtl* ctl, const ReduceArgs rE, const ReduceArgs rF, const int64_t alt_off)
{
    int64_t off = 0;
    if (ctl) {
        if (ctl->done)
            return;
        off = ctl->w_which ? 0 : alt_off;
    }
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool first = tid < 32 * rE.n_pose;
    if (!first)
        tid -= 32 * rE.n_pose;
    const ReduceArgs& r = first ? rE : rF;
    const int n_pose = r.n_pose;
    const int32_t* __restrict__ pose_task = r.pose_task;
    const double* __restrict__ part = r.part;
    double* __restrict__ Hout = r.Hout + off;
    double* __restrict__ gout = r.gout + off;
    double* __restrict__ pose_cost = r.pose_cost;
    const int p = tid >> 5, k = tid & 31;
    if (p >= n_pose || k >= 28)
        return;
    // four interleaved running sums, combined in a fixed order: the loads of a long task list (the fused kernel's
    // kept family: one partial per group of eliminated poses) do not wait for one another's additions
    const int t0 = pose_task ? pose_task[p] : 0, t1 = pose_task ? pose_task[p + 1] : r.n_fixed;
    const double* __restrict__ pp = part + (int64_t)k * r.sk + (int64_t)p * r.sp;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = t0;
    for (; t + 3 < t1; t += 4) {
        s0 += pp[(int64_t)t * r.st];
        s1 += pp[(int64_t)(t + 1) * r.st];
        s2 += pp[(int64_t)(t + 2) * r.st];
        s3 += pp[(int64_t)(t + 3) * r.st];
    }
    if (t < t1) s0 += pp[(int64_t)t * r.st];
    if (t + 1 < t1) s1 += pp[(int64_t)(t + 1) * r.st];
    if (t + 2 < t1) s2 += pp[(int64_t)(t + 2) * r.st];
    const double s = (s0 + s1) + (s2 + s3);
    if (k < 21) {
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= k)
            ++a;
        const int b = k - a * (a + 1) / 2;
        Hout[36 * (int64_t)p + 6 * a + b] = s;
        Hout[36 * (int64_t)p + 6 * b + a] = s;
    } else if (k < 27) {
        gout[6 * (int64_t)p + (k - 21)] = s;
    } else if (pose_cost) {
        pose_cost[p] = s;
    }
    if (r.gm_out) {
        // the six gradient components sit in lanes 21..26 of the pose's 32-lane group
        // (rows with p >= n_pose or k >= 28 left above: the shuffles below only read lanes that are here)
        double ng[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
            ng[j] = -__shfl(s, 21 + j, 32);
        if (k == 21) {
            const double* x = r.pose7 + 7 * (int64_t)p;
            double xp[7], gm = 0.0;
            if (r.euclid) {
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    xp[j] = x[j] + ng[j];
                xp[6] = x[6];
            } else {
                pose_plus(x, ng, xp);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j)
                gm = fmax(gm, fabs(x[j] - xp[j]));
            r.gm_out[p] = gm;
        }
    }
}

// out[0] = sum_{i<n} in[i*stride] in a fixed order (one block, pairwise tree over a serial prefix).
__global__ __launch_bounds__(256) void k_sum(const LmCtl* ctl, const double* __restrict__ in, int n, int stride,
                                             double* __restrict__ out)
{
    if (ctl && ctl->done)
        return;
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256)
        s += in[(int64_t)i * stride];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m)
            sh[threadIdx.x] += sh[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[0] = sh[0];
}

// Cost-only pass: 1/2 sum rho(|r|^2) per task (Ceres Evaluator with jacobians == NULL).
struct CostArgs {
    Intrinsics K;
    const Task* tasks;
    int n_tasks;
    const int32_t* other;
    const double* px;
    int64_t n_pad;
    const double* own_pose;
    const double* other_pose;
    const double* tag_wh;
    int robustify;
    double huber_a;
    double* part;  // [n_tasks]
    const int32_t* caller;
    const uint8_t* mask;
    const LmCtl* ctl;
};

template <bool OWN_IS_CAM, bool POINTS = false>
__global__ __launch_bounds__(256) void k_cost(const CostArgs a)
{
    if (a.ctl && a.ctl->done)
        return;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= a.n_tasks)
        return;
    const Task t = a.tasks[wave];
    const int64_t i = (int64_t)t.begin + lane;
    const bool valid = i < t.end;
    const int64_t is = valid ? i : t.begin;
    const int o = a.other[is];
    const int tag_idx = OWN_IS_CAM ? o : t.pose;
    const int cam_idx = OWN_IS_CAM ? t.pose : o;
    const double* camq = (OWN_IS_CAM ? a.own_pose : a.other_pose) + 7 * (int64_t)cam_idx;
    const double* tagq = (OWN_IS_CAM ? a.other_pose : a.own_pose) + 7 * (int64_t)tag_idx;
    Rigid cam, tag;
    double pt[6];
    if (POINTS) {
        load_rigid<false>(camq, cam);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            pt[k] = tagq[k];
    } else {
        load_rigid<true>(camq, cam);
        load_rigid<true>(tagq, tag);
    }
    const double hw = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx], hh = POINTS ? 0.0 : 0.5 * a.tag_wh[2 * tag_idx + 1];
    double cost = 0.0;
#pragma unroll
    for (int c = 0; c < (POINTS ? 2 : 4); ++c) {
        const double sx = (c == 1 || c == 2) ? hw : -hw;
        const double sy = (c >= 2) ? hh : -hh;
        CornerEval e;
        if (POINTS)
            eval_point<false, false>(a.K, cam, camq, pt[3 * c], pt[3 * c + 1], pt[3 * c + 2],
                                     a.px[(2 * c) * a.n_pad + is], a.px[(2 * c + 1) * a.n_pad + is], e);
        else
            eval_corner<false, false>(a.K, cam, tag, sx, sy, a.px[(2 * c) * a.n_pad + is],
                                      a.px[(2 * c + 1) * a.n_pad + is], e);
        double rho0, wgt;
        huber(a.robustify != 0, a.huber_a, e.ru * e.ru + e.rv * e.rv, rho0, wgt);
        cost += 0.5 * rho0;
    }
    cost = wave_sum((valid && a.mask[a.caller[is]]) ? cost : 0.0);
    if (lane == 0)
        a.part[wave] = cost;
}

// Reprojection statistics (src/TagReconstructor.cpp:340-455): per task the sum over corners of
// |projection - observation|_2, and optionally the signed per-corner errors in caller order.
// Rotation matrices come from the un-normalised quaternion, like Eigen's toRotationMatrix there.
struct StatsArgs {
    Intrinsics K;
    const Task* tasks;
    int n_tasks;
    const int32_t* other;
    const int32_t* caller;
    const double* px;
    int64_t n_pad;
    const double* own_pose;
    const double* other_pose;
    const double* tag_wh;
    double* part;        // [n_tasks]
    int32_t* part_n;     // [n_tasks] active observations of the task
    double* per_corner;  // [8*n_obs] caller order, or null (zeros for switched-off observations)
    const uint8_t* mask; // [n_obs] caller order
};

template <bool OWN_IS_CAM>
__global__ __launch_bounds__(256) void k_stats(const StatsArgs a)
{
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= a.n_tasks)
        return;
    const Task t = a.tasks[wave];
    const int64_t i = (int64_t)t.begin + lane;
    const bool valid = i < t.end;
    const int64_t is = valid ? i : t.begin;
    const int o = a.other[is];
    Rigid own, oth;
    load_rigid<false>(a.own_pose + 7 * (int64_t)t.pose, own);
    load_rigid<false>(a.other_pose + 7 * (int64_t)o, oth);
    const Rigid& cam = OWN_IS_CAM ? own : oth;
    const Rigid& tag = OWN_IS_CAM ? oth : own;
    const int tag_idx = OWN_IS_CAM ? o : t.pose;
    const double hw = 0.5 * a.tag_wh[2 * tag_idx], hh = 0.5 * a.tag_wh[2 * tag_idx + 1];
    const int64_t ci = a.caller[is];
    const bool on = valid && a.mask[ci];
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double sx = (c == 1 || c == 2) ? hw : -hw;
        const double sy = (c >= 2) ? hh : -hh;
        CornerEval e;
        // camModel.projectPoint(R * pts3d[i] + t) - tagObs.corners[i]   (src/TagReconstructor.cpp:361-362)
        eval_corner<false, false, true>(a.K, cam, tag, sx, sy, a.px[(2 * c) * a.n_pad + is],
                                        a.px[(2 * c + 1) * a.n_pad + is], e);
        sum += sqrt(e.ru * e.ru + e.rv * e.rv);
        if (a.per_corner && valid) {
            a.per_corner[8 * ci + 2 * c] = on ? e.ru : 0.0;
            a.per_corner[8 * ci + 2 * c + 1] = on ? e.rv : 0.0;
        }
    }
    sum = wave_sum(on ? sum : 0.0);
    const int n_on = __popcll(__ballot(on));
    if (lane == 0) {
        a.part[wave] = sum;
        a.part_n[wave] = n_on;
    }
}

// Per pose: the sum of its tasks' partial sums (task order) and the number of corner blocks behind it
// (4 per active observation): the numerators and denominators of src/TagReconstructor.cpp:366-373,412-422.
__global__ void k_stats_pose(int n_cam, const int32_t* __restrict__ task_c, const double* __restrict__ part_c,
                             const int32_t* __restrict__ cnt_c, int n_tag, const int32_t* __restrict__ task_t,
                             const double* __restrict__ part_t, const int32_t* __restrict__ cnt_t,
                             double* __restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_cam + n_tag)
        return;
    const bool is_cam = p < n_cam;
    const int q = is_cam ? p : p - n_cam;
    const int32_t* pt = is_cam ? task_c : task_t;
    const double* part = is_cam ? part_c : part_t;
    const int32_t* cnt = is_cam ? cnt_c : cnt_t;
    double s = 0.0;
    int64_t n = 0;
    for (int t = pt[q]; t < pt[q + 1]; ++t) {
        s += part[t];
        n += 4 * (int64_t)cnt[t];
    }
    out[p] = s;
    out[n_cam + n_tag + p] = (double)n;
}

// CameraModel::projectPoint (src/CameraModel.cpp:6-26) for a batch of camera-frame points.
__global__ void k_project(Intrinsics K, int64_t n, const double* __restrict__ pc, double* __restrict__ uv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double X = pc[3 * i], Y = pc[3 * i + 1], Z = pc[3 * i + 2];
    const double x = X / Z, y = Y / Z;
    const double r2 = x * x + y * y;
    const double rad = 1.0 + r2 * (K.k1 + r2 * (K.k2 + r2 * K.k3));
    double xd, yd;
    distort(K, true, x, y, r2, rad, xd, yd);   // :20-23: the y term uses the already distorted x
    uv[2 * i] = K.fx * xd + K.cx;
    uv[2 * i + 1] = K.fy * yd + K.cy;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

static inline int blocks_for_tasks(int n_tasks) { return (n_tasks + 3) / 4; }

// lm: the evaluation of the LM loop -- at the candidate poses, W into the buffer LmCtl::w_which does not name
static EvalArgs make_eval_args(Engine& e, const ObsOrder& ord, bool own_is_cam, void* W, bool lm = false)
{
    EvalArgs a;
    a.K = e.K;
    a.tasks = ord.tasks;
    a.n_tasks = ord.n_tasks;
    a.other = ord.other;
    a.px = ord.px;
    a.n_pad = ord.n_pad;
    const double* cam = lm ? e.cam_cand : e.cam_qt;
    const double* tag = lm ? e.tag_cand : e.tag_qt;
    a.own_pose = own_is_cam ? cam : tag;
    a.other_pose = own_is_cam ? tag : cam;
    a.tag_wh = e.tag_wh;
    a.fixed_tag = e.fixed_tag;
    a.fixed_shift = e.points ? 1 : 0;
    a.robustify = 0;
    a.huber_a = 1.0;
    a.part = ord.part;
    a.W = W;
    a.W_alt = (lm && W) ? (e.f32_accum ? (void*)e.Wf2 : (void*)e.W2) : nullptr;
    a.caller = ord.caller;
    a.mask = e.obs_mask;
    a.ctl = e.ctl;
    return a;
}

// use_ctl: the evaluation of an LM iteration -- at the CANDIDATE poses, blocks into the staging copy of the small
// buffer (they replace the working copy when the step is accepted), W into the buffer that does not hold W at x.
// Otherwise (covariance, vmm_ba_eval_blocks, kernel timing): at the current poses, into the working copies.
void launch_eval_passes(Engine& e, int robustify, double huber_a, bool use_ctl)
{
    const bool e_is_cam = e.elim_cams;
    const bool lm = use_ctl;
    EvalArgs aE = make_eval_args(e, e.ordE, e_is_cam, e.f32_accum ? (void*)e.Wf : (void*)e.W, lm);
    EvalArgs aF = make_eval_args(e, e.ordF, !e_is_cam, nullptr, lm);
    aE.robustify = aF.robustify = robustify;
    aE.huber_a = aF.huber_a = huber_a;
    if (!use_ctl)
        aE.ctl = aF.ctl = nullptr;
    const int nb_e = blocks_for_tasks(aE.n_tasks), nb_f = blocks_for_tasks(aF.n_tasks);
    if (e.fused_eval) {
        FusedArgs fa;
        fa.pair_obs = e.pair_obs;
        fa.e_list = e.fused_e_list;
        fa.e_part0 = e.fused_e_part0;
        fa.n_e_act = e.fused_n_e_act;
        fa.n_f = e.n_f;
        fa.n_f_pad = e.fused_f_pad;
        fa.n_chunks = e.fused_chunks;
        fa.group = e.fused_group;
        fa.n_groups = e.fused_groups;
        fa.partF = e.fused_partF;
        aE.part = e.fused_partE;
        const int n_waves = fa.n_groups * fa.n_chunks;
        const dim3 grid((n_waves + 3) / 4);
        if (n_waves > 0) {
            if (e.points) {
                if (e_is_cam)
                    hipLaunchKernelGGL((k_eval_fused<true, double, true>), grid, dim3(256), 0, e.stream, aE, fa);
                else
                    hipLaunchKernelGGL((k_eval_fused<false, double, true>), grid, dim3(256), 0, e.stream, aE, fa);
            } else if (e.f32_accum) {
                if (e_is_cam)
                    hipLaunchKernelGGL((k_eval_fused<true, float, false>), grid, dim3(256), 0, e.stream, aE, fa);
                else
                    hipLaunchKernelGGL((k_eval_fused<false, float, false>), grid, dim3(256), 0, e.stream, aE, fa);
            } else if (e_is_cam)
                hipLaunchKernelGGL((k_eval_fused<true, double, false>), grid, dim3(256), 0, e.stream, aE, fa);
            else
                hipLaunchKernelGGL((k_eval_fused<false, double, false>), grid, dim3(256), 0, e.stream, aE, fa);
        }
    } else if (nb_e + nb_f > 0) {
        if (e.points) {
            if (e_is_cam)
                hipLaunchKernelGGL((k_eval_both<true, double, true>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
            else
                hipLaunchKernelGGL((k_eval_both<false, double, true>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
        } else if (e.f32_accum) {
            if (e_is_cam)
                hipLaunchKernelGGL((k_eval_both<true, float>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
            else
                hipLaunchKernelGGL((k_eval_both<false, float>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
        } else if (e_is_cam)
            hipLaunchKernelGGL((k_eval_both<true>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
        else
            hipLaunchKernelGGL((k_eval_both<false>), dim3(nb_e + nb_f), dim3(256), 0, e.stream, aE, aF, nb_e);
    }
    const LmCtl* ctl = use_ctl ? e.ctl : nullptr;
    // LM loop, world > 1: the staging copy (all-reduced next); one GPU: the copy w_which does not name
    const bool stage = lm && e.multi;
    double* const oH_cam = stage ? e.ev_H_cam : e.H_cam;
    double* const oH_tag = stage ? e.ev_H_tag : e.H_tag;
    double* const og_cam = stage ? e.ev_g_cam : e.g_cam;
    double* const og_tag = stage ? e.ev_g_tag : e.g_tag;
    ReduceArgs rE, rF;
    rE.n_pose = e.n_e;
    rE.pose_task = e.ordE.pose_task;
    rE.part = e.ordE.part;
    rE.Hout = e_is_cam ? oH_cam : oH_tag;
    rE.gout = e_is_cam ? og_cam : og_tag;
    // per-pose cost of the eliminated family; an LM evaluation with world > 1 leaves it in the all-reduced staging buffer
    rE.pose_cost = (lm && e.multi) ? e.ev_pose_cost : e.part_cost;
    const bool want_gm = lm && !e.multi;   // world > 1: the gradient is only complete behind the all-reduce
    rE.pose7 = want_gm ? (e_is_cam ? e.cam_cand : e.tag_cand) : nullptr;
    rE.euclid = (e.points && !e_is_cam) ? 1 : 0;
    rE.gm_out = want_gm ? e.pose_gm + (e_is_cam ? 0 : e.n_cams) : nullptr;
    rF.pose7 = want_gm ? (e_is_cam ? e.tag_cand : e.cam_cand) : nullptr;
    rF.euclid = (e.points && e_is_cam) ? 1 : 0;
    rF.gm_out = want_gm ? e.pose_gm + (e_is_cam ? e.n_cams : 0) : nullptr;
    rF.n_pose = e.n_f;
    rF.pose_task = e.ordF.pose_task;
    rF.part = e.ordF.part;
    rE.n_fixed = rF.n_fixed = 0;
    rE.st = rF.st = kPart;
    rE.sk = rF.sk = 1;
    rE.sp = rF.sp = 0;
    if (e.fused_eval) {
        rE.pose_task = e.fused_pose_task;
        rE.part = e.fused_partE;
        rF.pose_task = nullptr;           // every kept pose: one partial per group of eliminated poses
        rF.n_fixed = e.fused_groups;
        rF.part = e.fused_partF;
        rF.st = 28 * (int64_t)e.fused_f_pad;
        rF.sk = e.fused_f_pad;
        rF.sp = 1;
    }
    rF.Hout = e_is_cam ? oH_tag : oH_cam;
    rF.gout = e_is_cam ? og_tag : og_cam;
    rF.pose_cost = nullptr;
    hipLaunchKernelGGL(k_reduce_pose, dim3(((e.n_e + e.n_f) * 32 + 255) / 256), dim3(256), 0, e.stream, ctl, rE, rF,
                       (lm && !e.multi) ? e.small_alt_off : (int64_t)0);
    if (!lm)   // LM evaluations: k_control sums the per-pose costs
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, e.stream, ctl, e.part_cost, e.n_e, 1,
                           e.cost_slot);
}

void launch_sum(Engine& e, bool guard, const double* in, int n, double* out)
{
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, e.stream, guard ? e.ctl : (const LmCtl*)nullptr, in, n, 1, out);
}

void launch_cost_kernel(Engine& e, const double* cam, const double* tag, bool guard, int robustify, double huber_a)
{
    CostArgs a;
    a.K = e.K;
    a.tasks = e.ordE.tasks;
    a.n_tasks = e.ordE.n_tasks;
    a.other = e.ordE.other;
    a.px = e.ordE.px;
    a.n_pad = e.ordE.n_pad;
    a.own_pose = e.elim_cams ? cam : tag;
    a.other_pose = e.elim_cams ? tag : cam;
    a.tag_wh = e.tag_wh;
    a.robustify = robustify;
    a.huber_a = huber_a;
    a.part = e.part_k1;
    a.caller = e.ordE.caller;
    a.mask = e.obs_mask;
    a.ctl = guard ? e.ctl : nullptr;
    if (a.n_tasks <= 0)
        return;
    const dim3 grid(blocks_for_tasks(a.n_tasks));
    if (e.points) {
        if (e.elim_cams)
            hipLaunchKernelGGL((k_cost<true, true>), grid, dim3(256), 0, e.stream, a);
        else
            hipLaunchKernelGGL((k_cost<false, true>), grid, dim3(256), 0, e.stream, a);
    } else if (e.elim_cams)
        hipLaunchKernelGGL((k_cost<true>), grid, dim3(256), 0, e.stream, a);
    else
        hipLaunchKernelGGL((k_cost<false>), grid, dim3(256), 0, e.stream, a);
}

void launch_cost(Engine& e, const double* cam, const double* tag, bool guard, int robustify, double huber_a,
                 double* out_scalar)
{
    launch_cost_kernel(e, cam, tag, guard, robustify, huber_a);
    launch_sum(e, guard, e.part_k1, e.ordE.n_tasks, out_scalar);
}

// Both statistics passes + the per-pose sums: e.stats_pose = [sum per camera | sum per tag | count per camera |
// count per tag] (counts as doubles).  All buffers belong to the handle.
void launch_stats(Engine& e, double* per_corner_dev)
{
    const ObsOrder& oc = e.elim_cams ? e.ordE : e.ordF;  // sorted by camera
    const ObsOrder& ot = e.elim_cams ? e.ordF : e.ordE;  // sorted by tag
    double* part_cam = e.stats_part;
    double* part_tag = e.stats_part + oc.n_tasks;
    int32_t* n_cam = e.stats_cnt;
    int32_t* n_tag = e.stats_cnt + oc.n_tasks;
    StatsArgs a;
    a.K = e.K;
    a.tag_wh = e.tag_wh;
    a.mask = e.obs_mask;
    a.tasks = oc.tasks; a.n_tasks = oc.n_tasks; a.other = oc.other; a.caller = oc.caller; a.px = oc.px;
    a.n_pad = oc.n_pad; a.own_pose = e.cam_qt; a.other_pose = e.tag_qt; a.part = part_cam; a.part_n = n_cam;
    a.per_corner = per_corner_dev;
    if (a.n_tasks > 0)
        hipLaunchKernelGGL((k_stats<true>), dim3(blocks_for_tasks(a.n_tasks)), dim3(256), 0, e.stream, a);
    a.tasks = ot.tasks; a.n_tasks = ot.n_tasks; a.other = ot.other; a.caller = ot.caller; a.px = ot.px;
    a.n_pad = ot.n_pad; a.own_pose = e.tag_qt; a.other_pose = e.cam_qt; a.part = part_tag; a.part_n = n_tag;
    a.per_corner = nullptr;
    if (a.n_tasks > 0)
        hipLaunchKernelGGL((k_stats<false>), dim3(blocks_for_tasks(a.n_tasks)), dim3(256), 0, e.stream, a);
    hipLaunchKernelGGL(k_stats_pose, dim3((e.n_cams + e.n_tags + 255) / 256), dim3(256), 0, e.stream, e.n_cams,
                       (const int32_t*)oc.pose_task, (const double*)part_cam, (const int32_t*)n_cam, e.n_tags,
                       (const int32_t*)ot.pose_task, (const double*)part_tag, (const int32_t*)n_tag, e.stats_pose);
}

void launch_project(hipStream_t st, const Intrinsics& K, int64_t n, const double* pc, double* uv)
{
    if (n > 0)
        hipLaunchKernelGGL(k_project, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, K, n, pc, uv);
}

// Touches every kernel of this file once (vmm_ba_create): the code object is loaded and the kernel's resources
// are known before any launch is recorded into a hipGraph (nothing may be loaded lazily under stream capture).
int preload_eval_kernels()
{
    hipFuncAttributes at;
    int bad = 0;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<true, double>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<false, double>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<true, float>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<false, float>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<true, double, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_both<false, double, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<true, double, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<false, double, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<true, float, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<false, float, false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<true, double, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_eval_fused<false, double, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cost<true, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cost<false, true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_reduce_pose)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_sum)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cost<true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_cost<false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_stats<true>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_stats<false>)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_stats_pose)) != hipSuccess;
    bad += hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_project)) != hipSuccess;
    return bad;
}

} // namespace vmm
