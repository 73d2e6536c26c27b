// Internal layout of the bundle-adjustment engine (host + device views).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/vmm_ba.h"
#include "geom.hpp"

namespace vmm {

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kPart = 32;          // doubles per task partial: 21 (H lower) + 6 (g) + 1 (cost) + pad
constexpr int kNB = 64;            // dense block size of the reduced system
constexpr int kDfMaskWords = 4;    // 64-bit words of a block row's structure mask (tree orderings): up to 255 block columns
constexpr int kDfMaxBlk = 256;
constexpr int kKT = 16;            // K tile of the MFMA f64 rank-k update (rows of Z per LDS stage)
constexpr int kST = 128;           // output tile of the rank-k update (the leading dimension is a multiple of it)
constexpr int kLdsRow = 80;        // LDS row stride (doubles) for 64-wide tiles: rows k, k+1 land in
                                   // opposite 32-bank halves for ds_read_b64 (MI355X_MICROARCH LDS)

// One wave's work: up to 64 consecutive observations of one pose in a family-sorted order.
struct Task {
    int32_t pose;
    int32_t begin;
    int32_t end;
};

// Observations sorted by one pose family ("own"); SoA so that lane = observation is coalesced.
struct ObsOrder {
    int64_t n = 0;          // observations
    int64_t n_pad = 0;      // SoA stride (multiple of 64)
    int32_t* own = nullptr;     // [n] pose index in the sorted family
    int32_t* other = nullptr;   // [n] pose index in the other family
    int32_t* caller = nullptr;  // [n] position in the caller's observation order
    double* px = nullptr;       // [8][n_pad]
    Task* tasks = nullptr;
    int32_t n_tasks = 0;
    int32_t* pose_task = nullptr;  // [n_pose+1] task range per pose
    double* part = nullptr;        // [n_tasks][kPart]
    int32_t* start = nullptr;      // [n_pose+1] observation range per pose
};

// Device control block of the trust-region loop (Ceres TrustRegionMinimizer +
// LevenbergMarquardtStrategy state).  Lives in device memory; the host only polls it.
struct LmCtl {
    // options (copied at solve start)
    int32_t max_num_iterations, robustify, jacobi_scaling, max_invalid;
    double huber_a, function_tolerance, gradient_tolerance, parameter_tolerance;
    double max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    // strategy
    double radius, decrease_factor;
    int32_t reuse_diagonal, num_invalid;
    // minimizer
    double x_cost, cand_cost, model_cost_change, x_norm, initial_cost;
    int32_t iteration;         // index of the iteration record being built
    int32_t first_eval;        // the pending evaluation is iteration zero
    int32_t done, termination; // done: 0 = running, 1 = the loop is over, 2 = paused: this pass's factorisation gave up
                               // waiting on another workgroup (sync_timeout), the host redoes it on the launch-per-
                               // block-column path and resumes (every kernel of a pass returns at once while done != 0)
    int32_t lin_fail;          // a Cholesky pivot was not positive in this iteration (numerical failures ONLY)
    // Spin give-ups of the two one-launch kernels that hand data between workgroups.  They are NOT numerical
    // failures: the trust-region policy never sees them.  sync_timeout: bit 0 k_chol_dataflow, bit 1
    // k_backsolve_chain, bit 2 another rank reported one (world > 1) -- of the pass that paused; the host clears it.
    int32_t sync_timeout;
    int32_t num_sync_timeouts; // passes of this solve redone on the fallback path
    int32_t sync_kernels;      // OR of sync_timeout over the solve
    uint32_t spin_limit_df, spin_limit_chain;   // 0 = the kernels' defaults (VMM_BA_DEBUG_SPIN_LIMIT shrinks them)
    int32_t spin_wg;           // debugging: the shrunk limit applies to this workgroup (blockIdx.x) only; < 0: to all
    int32_t records;           // trace rows pushed (== Ceres summary.iterations.size())
    int32_t num_successful, num_unsuccessful, num_lm_iterations, num_jac_evals, num_cost_evals;
    int32_t trace_capacity;
    int32_t w_which;           // which copy of W and of the small blocks (H, g) belongs to x; the evaluation at the
                               // candidate writes the other one and an accepted step flips this
    vmm_ba_iteration cur;      // record under construction
    // phase report (vmm_ba_summary.time_*_s): s_memrealtime (100 MHz) stamps written by thread 0 of the first
    // kernel of each group -- 0 evaluation at the candidate, 1 unused, 2 k_form_z, 3 Cholesky, 4 k_backsub, 5 k_control --
    // and their differences accumulated per solve: 0 evaluation, 1 control, 2 eliminate + rank-k update,
    // 3 factor + triangular solves, 4 step (back-substitution, candidate, cost at the candidate)
    unsigned long long stamp[6];
    unsigned long long phase_ticks[5];
};

// Offset (doubles) of the copy of the small blocks that belongs to x.
__device__ __forceinline__ int64_t small_sel(const LmCtl* ctl, const int64_t alt_off) { return ctl->w_which ? alt_off : 0; }

// Start-of-group stamp by one thread of the launch (costs one s_memrealtime + one 8-byte store).
__device__ __forceinline__ void phase_stamp(const LmCtl* ctl, int slot)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        const_cast<LmCtl*>(ctl)->stamp[slot] = __builtin_amdgcn_s_memrealtime();
}

// Stream-K decomposition of the rank-k update (kernels_schur.hip): device arrays + sizes.
struct SyrkPlan {
    int n_tiles = 0, n_kt = 0, n_wg = 0, n_segments = 0;
    int32_t *tile_bi = nullptr, *tile_bj = nullptr, *wg_seg0 = nullptr, *tile_seg0 = nullptr;
    int64_t *wg_u0 = nullptr, *wg_u1 = nullptr;   // [n_wg] unit range of each workgroup (by blockIdx)
    double* partials = nullptr;   // [n_segments][kST*kST]
    bool wide = false;            // k_syrk_wide: one 8-wave workgroup per CU, one tile per workgroup (few tiles)
};

struct Engine {
    int device = 0;
    int n_cu = 256;                 // compute units of the device (workgroup counts of the persistent loops)
    hipStream_t stream = nullptr;
    int rank = 0, world = 1;
    bool multi = false;             // world > 1 (or forced for tests): staging buffers, eager launches, all-reduces
    vmm_ba_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;      // ncclComm_t of vmm_ba_enable_rccl: the all-reduces are ncclAllReduce on `stream`
    bool rccl_graph = true;         // ... recorded into the iteration's hipGraph (VMM_BA_RCCL_GRAPH=0: enqueued between graphs)

    Intrinsics K;
    int n_cams = 0, n_tags = 0, fixed_tag = -1;
    // point landmarks (VMM_BA_LANDMARK_POINTS): the "tag" family holds 2 * n_tags_user point pairs, the observation
    // arrays 2 * n_obs_user pair observations; fixed_tag stays the caller's tag index (block >> 1 is compared)
    bool points = false;
    int n_tags_user = 0;
    int64_t n_obs_user = 0;
    std::vector<double> user_tag_wh;   // [2 * n_tags_user]: vmm_ba_set_state turns tag poses into corners again
    int64_t n_obs = 0;
    bool elim_cams = true;          // eliminated family E = cameras (else tags)
    int n_e = 0, n_f = 0;           // pose counts of the eliminated / kept family

    // poses: cameras then tags, 7 doubles each
    double* cam_qt = nullptr;
    double* tag_qt = nullptr;
    double* cam_cand = nullptr;
    double* tag_cand = nullptr;
    double* tag_wh = nullptr;

    ObsOrder ordE, ordF;            // observations sorted by the eliminated / kept family

    // normal-equation blocks ("small" buffer, contiguous for one all-reduce):
    //   H_cam[36*n_cams] | H_tag[36*n_tags] | g_cam[6*n_cams] | g_tag[6*n_tags] | cost | pad
    double* small = nullptr;
    size_t small_count = 0;
    double *H_cam = nullptr, *H_tag = nullptr, *g_cam = nullptr, *g_tag = nullptr, *cost_slot = nullptr;
    // where the evaluation kernels write (== the pointers above on one GPU, a staging copy that is
    // all-reduced first when world > 1)
    double* small_stage = nullptr;
    // one GPU: the staging copy IS the second copy of the small blocks (offset from `small` in doubles; kernels add it
    // when LmCtl::w_which is set); world > 1: 0 -- the all-reduced staging copy is copied into `small` on acceptance
    int64_t small_alt_off = 0;
    double *ev_H_cam = nullptr, *ev_H_tag = nullptr, *ev_g_cam = nullptr, *ev_g_tag = nullptr, *ev_cost = nullptr;
    double* ev_pose_cost = nullptr;   // [n_e] behind the staging copy: per-pose costs of the evaluation (world > 1)
    double* W = nullptr;            // [36][ordE.n_pad]: J_e^T J_f per observation, E order (f64 precision)
    double* W2 = nullptr;           // second buffer: every LM iteration evaluates at the candidate, LmCtl::w_which says
                                    // which one belongs to x
    float* Wf2 = nullptr;
    uint8_t* obs_mask = nullptr;    // [n_obs] caller order, 1 = observation active (vmm_ba_set_observation_mask)
    float* Wf = nullptr;            // the same in f32 (VMM_BA_PRECISION_F32_ACCUM); exactly one of the two exists
    bool f32_accum = false;

    // Fused evaluation (k_eval_fused): one evaluation per observation, lane = kept pose, the wave walks `fused_group`
    // eliminated poses.  On when most (e, f) pairs are observed (VMM_BA_EVAL=fused|twopass overrides).
    bool fused_eval = false;
    int32_t* pair_obs = nullptr;      // [n_e][fused_f_pad]
    int32_t* fused_e_list = nullptr;  // [fused_n_e_act] eliminated poses with observations
    int32_t* fused_e_part0 = nullptr; // [n_e] first partial of each eliminated pose
    int32_t* fused_pose_task = nullptr;   // [n_e + 1] partial range per eliminated pose (k_reduce_pose)
    double* fused_partE = nullptr;    // [fused_n_e_act * fused_chunks][kPart]
    double* fused_partF = nullptr;    // [fused_groups][28][fused_f_pad]
    int fused_n_e_act = 0, fused_f_pad = 0, fused_chunks = 0, fused_group = 1, fused_groups = 0;

    // tangent-space vectors, cameras first then tags (6 each)
    double *scale = nullptr, *diag = nullptr, *D2 = nullptr, *delta = nullptr;
    int32_t* active = nullptr;      // per pose (cameras then tags)

    // elimination
    double* Le = nullptr;           // [n_e][36] Cholesky factors of the damped E blocks
    double* ze = nullptr;           // [n_e][6]  L_e^{-1} g_e
    double* Z = nullptr;            // [k_pad][ldz] dense L_e^{-1} W (+ z column), row-major
    int k_dim = 0, k_pad = 0;
    int n_red = 0, n_pad = 0, ldz = 0, n_blk = 0;   // reduced order, padded, leading dim, blocks
    // Block-sparse elimination (kernels_schur.hip: k_schur_pairs): Z holds only the 6x6 blocks of co-observed (e, f) pairs and
    // S -= Z^T Z runs over pairs that share an eliminated pose.  Chosen at create by a cost model (VMM_BA_SCHUR=
    // dense|sparse overrides): at full visibility the dense MFMA rank-k update is the faster one.
    bool sparse_schur = false;
    double* Zc = nullptr;           // [n_obs][36]: one column-major 6x6 block per observation, E order
    int32_t* f2e = nullptr;         // [n_obs] F-order position -> E-order position of the same observation
    // the symbolic structure of S -= Z^T Z (built once at create): pairs of kept poses (row f: f' = 0..f, then the rhs),
    // the terms of every pair, and the work items of k_schur_pairs (row | first pair; up to 42 pairs of one row each)
    int32_t* pair_start = nullptr;  // [n_f + 1]
    int32_t* pair_tstart = nullptr; // [n_pairs + 1]
    int32_t* pair_terms = nullptr;  // [n_terms][2] position of the left block in the row's observation list | E-order
                                    // index of the right block (rhs pair: the eliminated pose)
    int32_t* row_items = nullptr;   // [2][n_row_items]
    double co_terms = 0.0;          // sum over the eliminated poses of (observations)^2: size of the host's pair bookkeeping
    bool explicit_pairs = false;    // only the pairs that share an eliminated pose are listed (pair_col), S zero-filled first
    int32_t* pair_col = nullptr;    // [n_pairs] first column of the pair's block in S (-1: the row's right-hand side entry)
    int32_t* row_of = nullptr;      // [n_f] first row of every kept pose in the reduced system (explicit form only)
    int32_t* pose_of_row = nullptr; // [n_pad] world > 1 with a tree ordering: kept pose of a row of the reduced system, -1: padding
    std::vector<int32_t> h_row_of;  // host copy; empty: kept pose f sits at row 6 f
    unsigned long long* chol_nz = nullptr;    // [n_blk + 1][kDfMaskWords] block structure of the factor (tree ordering), see DfArgs::nz
    unsigned char* chol_order = nullptr;      // [n_blk][kDfMaxBlk] panel order per block column, see DfArgs::order
    int32_t* df_wg = nullptr;                 // [n_df_wg][2] (block column, block row) of the tree-ordered launch's workgroups
    int32_t* df_slot = nullptr;               // [n_blk][n_blk + 1] slot of a block's slices in df_gran, -1: structurally zero
    int n_df_wg = 0;
    size_t df_tree_slots = 0;                 // blocks with published slices under the tree ordering
    bool chol_nz_on = false;                  // off while a call factors the dense, naturally ordered system (covariance)
    std::vector<int32_t> nd_node_first_blk;   // tree ordering: first 64-row block of every node, in elimination order
    int n_row_items = 0;
    double schur_flops = 0.0;       // algorithmic flops of the reduced-system formation on the path in use
    double chol_flops = 0.0;        // tree-ordered factor: flops over its non-zero blocks (0: dense factor)
    SyrkPlan syrk;                  // stream-K plan of S = Z^T Z
    double* S = nullptr;            // [ldz][ldz] reduced system (lower) + rhs row at n_pad
    double* S_packed = nullptr;     // world > 1: rows 0..n_pad of the lower triangle, packed, for the all-reduce
    double* P = nullptr;            // [4][kNB][ldz] transposed Cholesky panels (ring: panel k in slot k & 3)
    double* P4[4] = { nullptr, nullptr, nullptr, nullptr };
    double* dinv = nullptr;         // [ldz] reciprocals of the Cholesky diagonal
    double* Ldiag = nullptr;        // [n_blk][64][64] Cholesky factors of the diagonal blocks
    double* Linv = nullptr;         // [n_blk][64][64] their inverses (all but the last block)
    unsigned* flags = nullptr;      // [256] unused | epoch word | abort word | two tile counters of k_chol_step | agreement word
    unsigned long long* gran = nullptr;   // [2 * ld] {epoch, 32 value bits} granules of the chain's hand-offs
    bool no_chain = false;          // VMM_BA_NO_CHAIN=1: per-block back-substitution kernels
    unsigned long long* df_gran = nullptr;   // published 64x8 slices of the dataflow factorisation (<= 21 blocks)
    double* df_compact = nullptr;            // the same blocks once they are COMPLETE, as plain doubles [slot][column][row]
    unsigned* df_done = nullptr;             // [slot] == factorisation epoch: the block's compact copy is written
    bool no_dataflow = false;       // VMM_BA_NO_DATAFLOW=1: one k_chol_step launch per block column
    // debugging: VMM_BA_DEBUG_SPIN_LIMIT=<polls> [VMM_BA_DEBUG_SPIN_KERNEL=df|chain|both] [VMM_BA_DEBUG_SPIN_ONCE=1]
    // force spin give-ups in the one-launch factorisation / back-substitution (tests/test_gpu_edge_cases.py)
    uint32_t dbg_spin_df = 0, dbg_spin_chain = 0;
    bool dbg_spin_once = false;
    int dbg_spin_wg = -1;           // VMM_BA_DEBUG_SPIN_WG=<blockIdx.x>: only that workgroup gives up
    double* yf = nullptr;           // [ldz] solution of the reduced system (scaled coordinates)
    double* step_comm = nullptr;    // [7*n_e + 1]: delta of the eliminated family | per-pose cross terms | votes
    double* cost_comm = nullptr;    // [2] candidate cost (all-reduced)
    double* part_cost = nullptr;    // [ordE.n_tasks]
    double* part_cross = nullptr;   // [ordE.n_tasks] cross-term wave partials
    double* part_k1 = nullptr;      // [ordE.n_tasks] candidate-cost wave partials
    double* pose_part = nullptr;    // [n_pose][5] per-pose terms of the step decision
    double* pose_gm = nullptr;      // [n_pose] per-pose gradient max-norm terms of the evaluation at the candidate

    // reprojection statistics (vmm_ba_reprojection_stats): per-task partials, per-pose sums | counts
    double* stats_part = nullptr;   // [n_tasks by camera + n_tasks by tag]
    int32_t* stats_cnt = nullptr;   // same layout: active observations per task
    double* stats_pose = nullptr;   // [2 * (n_cams + n_tags)]
    double* stats_corner = nullptr; // [8 * n_obs], allocated by the first call that asks for per-corner errors

    LmCtl* ctl = nullptr;           // device
    LmCtl* ctl_host = nullptr;      // pinned
    // vmm_ba_set_state: the caller's poses are staged here (pinned) and copied on the stream without waiting for it
    double* pose_stage = nullptr;   // [7 * (n_cams + n_tags)]
    hipEvent_t pose_ev = nullptr;   // recorded behind the copies: the staging buffer is free again
    bool pose_ev_pending = false;
    double* pose_stage_dev = nullptr;   // the staging buffer as the device sees it (k_begin_loop reads it directly)
    bool dirty_cam = false, dirty_tag = false;   // staged by vmm_ba_set_state, not yet on the device
    vmm_ba_iteration* trace = nullptr;  // device
    int trace_capacity = 0;

    // one LM iteration captured as a hipGraph (single GPU; collectives are host calls)
    hipGraphExec_t iter_graph = nullptr;
    hipGraphExec_t iter_graph_seg[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };   // world > 1: the groups between the all-reduces
    int graph_robustify = -1;
    double graph_huber_a = 0.0;
    bool use_graph = true;
    int graph_passes = 2;           // LM passes recorded into one iteration graph (VMM_BA_GRAPH_PASSES; 2 measured best)
    int last_passes = 1;            // passes the last run_iteration enqueued
    bool eager_first = false;       // VMM_BA_EAGER_FIRST=1: the handle's first iteration is enqueued without capture
    bool launched_eagerly = false;

    std::vector<void*> allocs;
};

void set_error(const std::string& s);

// one per .hip file: hipFuncGetAttributes on every kernel (returns the number of failures)
int preload_eval_kernels();
int preload_schur_kernels();
int preload_chol_kernels();
int preload_lm_kernels();
int preload_cov_kernels();

// ---- kernel launchers (defined in the .hip files) ----
// kernels_eval.hip
void launch_eval_passes(Engine& e, int robustify, double huber_a, bool use_ctl);
void launch_cost_kernel(Engine& e, const double* cam, const double* tag, bool guard, int robustify, double huber_a);
void launch_cost(Engine& e, const double* cam, const double* tag, bool guard, int robustify, double huber_a,
                 double* out_scalar);
void launch_stats(Engine& e, double* per_corner_dev);
void launch_project(hipStream_t st, const Intrinsics& K, int64_t n, const double* pc, double* uv);
void launch_sum(Engine& e, bool guard, const double* in, int n, double* out);
// kernels_schur.hip
void launch_elim(Engine& e);
void launch_schur_rows(Engine& e, bool add_diag);   // block-sparse: S from the pair lists (k_schur_pairs)
int schur_pairs_per_item();                           // pairs of one row a workgroup of k_schur_pairs takes
void launch_syrk_only(Engine& e);
void launch_syrk_reduced(Engine& e);
void launch_pack_lower(Engine& e, bool unpack);
void launch_syrk_plan(hipStream_t st, const LmCtl* ctl, const double* Z, int ldz, const SyrkPlan& p);
void launch_reduce_plan(hipStream_t st, const LmCtl* ctl, const SyrkPlan& p, int ld, int n_rows, double* S);
// kernels_chol.hip
// safe: the launch-per-block-column factorisation and the per-block back-substitution (no workgroup waits on another)
void launch_cholesky_solve(Engine& e, double* S, int n_pad, int ld, double* y, LmCtl* ctl, bool safe = false);
void launch_chol_inverse(Engine& e, int k);
int dataflow_max_workgroups(int n_cu);
int dataflow_blocks(int n_blk, int n_cu);
struct CholLaunch {   // one k_chol_step launch of the launch-per-column factorisation
    int k;            // block column its panel workgroups factor; -1: the update-only hand-over launch
    int lazy[2];      // panels those workgroups first apply to their own column (older first; -1: none)
    int upd[2];       // panels of the launch's trailing update (-1: none; both: rank-128)
    int c0, t0, t1;   // first block column of the update's tile list, and the range of that list this launch takes
};
std::vector<CholLaunch> chol_step_schedule(int n_blk, int n_df);
void chol_schedule_tile(int n_blk, const CholLaunch& L, int t, int* bi, int* bj);   // trailing block columns factored by k_chol_dataflow (n_blk: all; 0: none)
int dataflow_workgroups(int n_blk);   // workgroups of k_chol_dataflow (must all fit on the chip at one per CU)
// kernels_cov.hip
void launch_cov_prepare(Engine& e);
void launch_cov_rhs(Engine& e, double* B, int ldb, bool identity_rhs);
void launch_cov_trsm(Engine& e, double* B, int ldb, int n_chunks, bool identity_rhs);
void launch_cov_gram(Engine& e, const double* X, int ldb, double* cov_dev);
// kernels_lm.hip
void launch_control(Engine& e);
void launch_begin_loop(Engine& e, const LmCtl& init);
void launch_pose_plus(hipStream_t st, int64_t n, const double* qt, const double* delta, double* out);
void launch_backsub(Engine& e);
void launch_candidate(Engine& e);

} // namespace vmm
