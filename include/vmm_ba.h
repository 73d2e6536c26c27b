/*
 * vmm_ba.h -- C-ABI of libvmm_ba.so: the MI355X (gfx950) bundle-adjustment engine behind
 * visual_marker_mapping's TagReconstructor hot path.
 *
 * Plain C: pointers and sizes only, no Eigen/torch types.  Each entry point names the reference
 * interface it replaces (file:line relative to /root/reference).  Host arrays are owned by the
 * caller; the handle owns device memory; nothing is thrown across this boundary -- every call
 * returns a status and vmm_ba_last_error() holds the text of the last failure on this thread.
 *
 * Data conventions (reference: include/visual_marker_mapping/Camera.h:13-17,
 * TagReconstructor.h:19-52, DetectionResults.h:10-37):
 *   pose      = 7 doubles: quaternion (w,x,y,z), translation (x,y,z)
 *   camera    = world->camera, tag = tag->world (TagReconstructionCostFunction.h:107-122)
 *   tag quad  = LL,LR,UR,UL = (-w/2,-h/2,0),(w/2,-h/2,0),(w/2,h/2,0),(-w/2,h/2,0)
 *   obs_px    = 8 doubles per tag observation: the four corners' (u,v) in that order
 *   tangent   = 6 per pose: translation(3), then the half-angle rotation vector(3) of
 *               ceres::QuaternionParameterization (src/TagReconstructor.cpp:661)
 */
#ifndef VMM_BA_H_
#define VMM_BA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMM_BA_ABI_VERSION 5

typedef struct vmm_ba_handle_s* vmm_ba_handle;

/* status codes */
enum {
    VMM_BA_OK = 0,
    VMM_BA_ERR_ARGUMENT = 1,   /* bad sizes / indices / null pointers */
    VMM_BA_ERR_HIP = 2,        /* a HIP runtime call failed (no device, OOM, launch failure) */
    VMM_BA_ERR_COLLECTIVE = 3, /* the user-supplied all-reduce callback reported failure */
    VMM_BA_ERR_STATE = 4,      /* call sequence error */
    VMM_BA_ERR_NUMERIC = 5     /* rank-deficient Jacobian: no covariance (ceres::Covariance::Compute == false) */
};

/* which pose family is eliminated by block Gaussian elimination before the dense reduced solve.
 * The reference puts tags in Ceres ordering group 0 and cameras in group 1
 * (src/TagReconstructor.cpp:675-676,695-696) but keeps Ceres' default exact solver; any exact
 * elimination yields the same LM step. */
enum {
    VMM_BA_ELIM_AUTO = 0,    /* eliminate the larger family -> smaller reduced system */
    VMM_BA_ELIM_TAGS = 1,    /* reduced camera system (the reference's ordering) */
    VMM_BA_ELIM_CAMERAS = 2  /* reduced tag system */
};

/* Ceres termination_type as printed by src/TagReconstructor.cpp:740 */
enum { VMM_BA_CONVERGENCE = 0, VMM_BA_NO_CONVERGENCE = 1, VMM_BA_FAILURE = 2 };

/* The problem TagReconstructor::doBundleAdjustment assembles at src/TagReconstructor.cpp:646-724:
 * reconstructed tags, reconstructed cameras with >= 1 reconstructed tag, and the observations
 * whose camera and tag are both reconstructed, with dense 0-based indices. */
typedef struct vmm_ba_problem {
    double intr[4];          /* fx, fy, cx, cy           CameraModel.h:14-17 */
    double dist[5];          /* k1, k2, p1, p2, k3       CameraModel.cpp:11-16 */
    int32_t n_cams;
    int32_t n_tags;
    const double* cam_qt;    /* [7*n_cams] initial camera poses (src/TagReconstructor.cpp:692-693) */
    const double* tag_qt;    /* [7*n_tags] initial tag poses    (src/TagReconstructor.cpp:665-666) */
    const double* tag_wh;    /* [2*n_tags] tag width, height    (src/TagReconstructor.cpp:713,718) */
    int32_t fixed_tag;       /* dense index of the origin tag, or -1 (src/TagReconstructor.cpp:669-673) */
    int64_t n_obs;
    const int32_t* obs_cam;  /* [n_obs] */
    const int32_t* obs_tag;  /* [n_obs] */
    const double* obs_px;    /* [8*n_obs] (src/DetectionIO.cpp:45-51) */
} vmm_ba_problem;

#define VMM_BA_PRECISION_F64 0
#define VMM_BA_PRECISION_F32_ACCUM 1

/* Landmark model.  TAG_POSES: TagReconstructionCostFunction (CostFunction.h:88-184), one 7-parameter pose per tag
 * -- the live doBundleAdjustment (src/TagReconstructor.cpp:646-743).  POINTS: OpenCVReprojectionError
 * (CostFunction.h:9-84), four free 3-D points per tag, 3x3 landmark blocks -- doBundleAdjustment_points
 * (src/TagReconstructor.cpp:457-644, `#if 0` in the reference): the tag poses handed to vmm_ba_create become their
 * four world corners (:483-491), the origin tag's corners are constant (:494-497), no loss function (:556).
 * vmm_ba_get_state then returns tag poses rebuilt from the optimised corners (:608-639), vmm_ba_get_points the
 * corners themselves. */
#define VMM_BA_LANDMARK_TAG_POSES 0
#define VMM_BA_LANDMARK_POINTS 1

typedef struct vmm_ba_create_options {
    int32_t device;          /* HIP device ordinal */
    int32_t elimination;     /* VMM_BA_ELIM_* */
    /* Multi-GPU: one process per GPU.  Every rank passes ALL poses and only ITS shard of the
     * observations (sharded by the eliminated family, cameras by default); rank/world are
     * informational, the exchange itself goes through vmm_ba_set_allreduce(). */
    int32_t rank;
    int32_t world_size;
    /* VMM_BA_PRECISION_F64 (default): everything in f64, as the reference.
     * VMM_BA_PRECISION_F32_ACCUM (BASELINE.json configs[3]): the Gauss-Newton blocks J^T J (per-pose 6x6
     * and per-observation J_e^T J_f) are accumulated and stored in f32; residuals, cost, the gradient
     * J^T r, the reduced system S, its factorisation and every LM decision stay f64.  The fixed point
     * (zero gradient) is unchanged, the LM trajectory is that of a slightly perturbed Gauss-Newton model. */
    int32_t precision;
    int32_t landmarks;       /* VMM_BA_LANDMARK_* */
    /* world_size > 1 only, optional (ABI 4): the (camera, tag) index pairs of the observations of ALL ranks -- the same
     * arrays on every rank, structure only, no pixels; the caller keeps them alive during vmm_ba_create only.  The
     * reduced systems of the ranks are summed and must share one layout: with the global structure every rank orders the
     * kept family by the same nested dissection (tree ordering, vmm_ba_summary.tree_ordering); without it world_size > 1
     * keeps the natural order.  The reference is a single process (src/TagReconstructor.cpp:646-743): no counterpart. */
    int64_t n_structure_obs;
    const int32_t* structure_obs_cam;
    const int32_t* structure_obs_tag;
} vmm_ba_create_options;

/* Solver::Options fields the reference sets (src/TagReconstructor.cpp:725-735) plus the Ceres
 * defaults that are in force because it does not set them. */
typedef struct vmm_ba_options {
    int32_t max_num_iterations;        /* 400 / 1500: src/TagReconstructor.cpp:233,271,277 */
    int32_t robustify;                 /* HuberLoss(huber_a) per corner block, :721 */
    double huber_a;                    /* 1.0 */
    double function_tolerance;         /* 1e-6  */
    double gradient_tolerance;         /* 1e-10 */
    double parameter_tolerance;        /* 1e-8  */
    double initial_trust_region_radius;/* 1e4   */
    double max_trust_region_radius;    /* 1e16  */
    double min_trust_region_radius;    /* 1e-32 */
    double min_relative_decrease;      /* 1e-3  */
    double min_lm_diagonal;            /* 1e-6  */
    double max_lm_diagonal;            /* 1e32  */
    int32_t max_num_consecutive_invalid_steps; /* 5 */
    int32_t jacobi_scaling;            /* 1 */
    int32_t num_threads;               /* accepted for signature parity (:733); the GPU path ignores it */
    int32_t poll_interval;             /* iteration-graph launches (two LM passes each) enqueued between host
                                          polls of the device control block (>=1); does not change results */
} vmm_ba_options;

/* One row of Ceres' Solver::Summary::iterations. */
typedef struct vmm_ba_iteration {
    int32_t iteration;
    int32_t step_is_valid;
    int32_t step_is_successful;
    int32_t reserved;
    double cost;
    double cost_change;
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;
    double model_cost_change;
} vmm_ba_iteration;

typedef struct vmm_ba_summary {
    int32_t termination_type;       /* VMM_BA_CONVERGENCE / NO_CONVERGENCE / FAILURE */
    int32_t iterations;             /* == Ceres summary.iterations.size() */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t num_lm_iterations;      /* passes of the trust-region loop (each = 1 linear solve) */
    int32_t num_jacobian_evals;
    int32_t num_cost_evals;
    int32_t elimination;            /* the VMM_BA_ELIM_* actually used */
    double initial_cost;
    double final_cost;
    double time_solve_s;            /* host wall time of vmm_ba_solve */
    vmm_ba_iteration* trace;        /* optional caller buffer, filled up to trace_capacity rows */
    int32_t trace_capacity;
    int32_t reserved;
    /* Where the device time of the solve went (the per-phase part of Ceres' Summary::FullReport(),
     * src/TagReconstructor.cpp:741-742), measured ON the device: the first kernel of every group stamps the
     * 100 MHz s_memrealtime counter and the control kernel sums the differences.  Their sum is <= time_solve_s. */
    double time_eval_s;             /* residual + Jacobian evaluation, J^T J / J^T r blocks (Ceres: "Jacobian & residual evaluation") */
    double time_eliminate_s;        /* block elimination, Z, rank-k update, reduced system */
    double time_factor_solve_s;     /* dense Cholesky + triangular solves (Ceres: "Linear solver") */
    double time_step_s;             /* back-substitution, candidate, cost at the candidate */
    double time_control_s;          /* trust-region control kernels */
    /* The one-launch factorisation (k_chol_dataflow) and back-substitution (k_backsolve_chain) hand data between
     * workgroups with bounded spins.  A spin that gives up (a GPU time-sliced between processes, a profiler
     * serialising workgroups) is NOT a numerical failure and never reaches the trust-region policy: the library
     * redoes that pass's factorisation on the launch-per-block-column path and goes on, so the trajectory is the
     * one of an undisturbed run.  These two fields report that it happened. */
    int32_t num_sync_timeouts;      /* LM passes of this solve whose factorisation was redone on the fallback path */
    int32_t sync_timeout_kernels;   /* OR over those passes: 1 = k_chol_dataflow, 2 = k_backsolve_chain gave up,
                                       4 = another rank reported a give-up (world_size > 1) */
    int32_t block_sparse;           /* 1: the elimination ran over co-observed (camera, tag) pairs only (compressed Z,
                                       k_schur_pairs) -- what the handle chose at create from its block structure,
                                       VMM_BA_SCHUR=dense|sparse overrides; 0: dense Z + MFMA rank-k update */
    int32_t tree_ordering;          /* block-sparse handles only: number of nodes of the nested-dissection tree the kept
                                       family is ordered by (their block columns of the reduced system's factor are
                                       computed independently of each other where the tree says so); 0: natural order.
                                       Chosen at create when the longest chain of dependent block columns shrinks enough;
                                       VMM_BA_ORDER=nd|natural overrides */
} vmm_ba_summary;

/* Sum-all-reduce of `count` doubles in DEVICE memory, in place, ordered on `hip_stream`
 * (a hipStream_t).  Return 0 on success.  Called by vmm_ba_solve / vmm_ba_cost on every rank in
 * the same order.  The default (none set) is the single-GPU identity. */
typedef int (*vmm_ba_allreduce_fn)(void* user, void* device_buffer, size_t count, void* hip_stream);

/* Average duration of each device kernel of one LM iteration, measured with HIP events on the
 * engine's own stream (bench.py's roofline leg). */
typedef struct vmm_ba_kernel_times {
    double eval_elim_ms;     /* residual+Jacobian+accumulate as an LM iteration runs it: both family passes in
                              * one launch (k_eval_both, the eliminated family's pass writes W) + the per-pose sums */
    double eval_keep_ms;     /* 0 since the two passes share a launch (kept for layout compatibility) */
    double cost_ms;          /* cost-only residual pass */
    double form_z_ms;        /* block factor + Z = L^-1 W */
    double syrk_ms;          /* reduced system: S -= Z^T Z (dense: f64 MFMA rank-k update; block-sparse: k_schur_rows) */
    double cholesky_ms;      /* dense Cholesky + triangular solves of the reduced system */
    double backsub_ms;       /* back-substitution + Plus + model cost */
    double lm_iteration_ms;  /* one whole LM iteration as enqueued by vmm_ba_solve */
    int64_t n_obs;
    int32_t reduced_dim;     /* order of the dense reduced system (without padding) */
    int32_t elim_dim;        /* 6 * number of eliminated poses */
    int32_t schur_sparse;    /* 1: the reduced system is formed over co-observed (e, f) pairs only (compressed Z) */
    int32_t syrk_wide;       /* 1: the dense rank-k update runs k_syrk_wide (one 8-wave workgroup per CU; few tiles) */
    double schur_flops;      /* algorithmic flops of that formation: dense (n+1)(n+2) K; block-sparse 432 per pair of
                                observations sharing an eliminated pose (lower triangle) + the right-hand side */
    double chol_flops;       /* ABI 5.  Tree-ordered factor (vmm_ba_summary.tree_ordering > 0): flops of the Cholesky
                                factorisation + the two triangular solves over the NON-ZERO 64 x 64 blocks of the factor
                                (after fill); 0 for a dense factor, whose count is n^3 / 3 + 2 n^2 */
} vmm_ba_kernel_times;

const char* vmm_ba_last_error(void);
int vmm_ba_abi_version(void);
void vmm_ba_default_options(vmm_ba_options* o);
void vmm_ba_default_create_options(vmm_ba_create_options* o);

/* Replaces the ceres::Problem construction of src/TagReconstructor.cpp:657-724: uploads poses and
 * observations, sorts them by pose family, builds the block structure.  Done once per problem. */
int vmm_ba_create(const vmm_ba_problem* problem, const vmm_ba_create_options* copt,
                  vmm_ba_handle* out);
void vmm_ba_destroy(vmm_ba_handle h);

/* Poses live on the device between calls (the reference mutates map nodes in place through raw
 * double*, src/TagReconstructor.cpp:665-666,692-693,722).  vmm_ba_set_state copies the caller's arrays into
 * pinned staging memory and issues no device command: the caller's buffers are free at once, and the next call that
 * needs the poses on the device uploads them (vmm_ba_solve inside the one launch that starts its loop).  Either
 * pointer may be NULL (that family is left alone).
 * VMM_BA_LANDMARK_POINTS handles: get_state followed by set_state is NOT the identity on the tag family --
 * get_state returns poses rebuilt (and re-orthogonalised) from the optimised corners, set_state regenerates exact
 * rectangles from pose and tag_wh, as vmm_ba_create does (src/TagReconstructor.cpp:483-491 / :608-639); the free
 * corners themselves are read with vmm_ba_get_points. */
int vmm_ba_set_state(vmm_ba_handle h, const double* cam_qt, const double* tag_qt);
int vmm_ba_get_state(vmm_ba_handle h, double* cam_qt, double* tag_qt);

/* VMM_BA_LANDMARK_POINTS handles: points[12 * n_tags] = the four world corners (LL, LR, UR, UL) of every tag,
 * the parameter blocks of doBundleAdjustment_points (src/TagReconstructor.cpp:485-492). */
int vmm_ba_get_points(vmm_ba_handle h, double* points);

int vmm_ba_set_allreduce(vmm_ba_handle h, vmm_ba_allreduce_fn fn, void* user);

/* Native collective path (BASELINE.json north_star: "RCCL all-reduce over xGMI of the reduced camera system"):
 * the library resolves librccl.so itself (dlopen) and issues ncclAllReduce(ncclDouble, ncclSum) in place on its
 * own stream, recorded into the LM iteration's hipGraph, so that world > 1 runs one graph per iteration with no
 * host callback.  Rank 0 draws an id, the host side hands the same 128 bytes to every rank by whatever means it
 * has (MPI, torch.distributed, a file), and every rank calls vmm_ba_enable_rccl -- a collective call; rank and
 * world size are those of vmm_ba_create_options.  Takes precedence over a vmm_ba_set_allreduce callback. */
#define VMM_BA_RCCL_ID_BYTES 128
/* 1 when librccl.so and every entry point the library uses were resolved in this process, else 0 (no device call).
 * Launchers ask every rank for this and agree on the answer BEFORE any rank enters vmm_ba_enable_rccl: a rank that
 * cannot load RCCL would otherwise leave the others blocked inside ncclCommInitRank. */
int vmm_ba_rccl_available(void);
int vmm_ba_rccl_unique_id(void* id128);
int vmm_ba_enable_rccl(vmm_ba_handle h, const void* id128);

/* Switches observations off and on without rebuilding the handle: mask[i] != 0 keeps observation i (the
 * caller's order), NULL keeps all.  This is how the incremental driver (src/TagReconstructor.cpp:86-278: one
 * more image per bundle adjustment, observations of unreconstructed tags skipped at :699-708) grows its problem
 * on the device: one handle for the whole detection set, a mask per step.  A pose left without an active
 * observation drops out of the reduced program exactly like a pose without observations.  The poses of
 * switched-off observations must still be finite numbers (they are evaluated and weighted 0). */
int vmm_ba_set_observation_mask(vmm_ba_handle h, const uint8_t* mask);

/* Replaces ceres::Solve at src/TagReconstructor.cpp:737-738. */
int vmm_ba_solve(vmm_ba_handle h, const vmm_ba_options* opt, vmm_ba_summary* summary);

/* Cost-only evaluation 1/2 sum rho(|r|^2) at the current state (Ceres Evaluator, cost only). */
int vmm_ba_cost(vmm_ba_handle h, int robustify, double huber_a, double* cost);

/* Replaces computeReprojectionErrorPerImg / PerTag / PerCorner (src/TagReconstructor.cpp:340-455):
 * per_cam_mean[n_cams] (-1 for a camera without observations, :379-383), per_tag_mean[n_tags]
 * (NaN for a tag without observations), *avg (:416-426), per_corner[8*n_obs] signed pixel errors in
 * the caller's observation order (:447-451).  Any output may be NULL. */
int vmm_ba_reprojection_stats(vmm_ba_handle h, double* per_cam_mean, double* per_tag_mean,
                              double* avg, double* per_corner);

/* Replaces the ceres::Covariance block of doBundleAdjustment (src/TagReconstructor.cpp:744-783):
 * cov[9*t .. 9*t+8] = row-major 3x3 covariance of tag t's translation = the corresponding block of
 * (J^T J)^-1 in tangent coordinates at the current state, J with the loss applied when robustify != 0
 * (Covariance::Options::apply_loss_function defaults to true).  Constant (origin) and residual-free tags
 * get zeros, as Ceres reports for constant blocks.  Computed from the Schur factor of the undamped,
 * unscaled normal equations; VMM_BA_ERR_NUMERIC if they are not positive definite.  Single-GPU handles. */
int vmm_ba_tag_translation_covariance(vmm_ba_handle h, int robustify, double huber_a, double* cov);

/* Replaces CameraModel::projectPoint (src/CameraModel.cpp:6-26) for n camera-frame points. */
int vmm_ba_project_points(const double intr[4], const double dist[5], int64_t n,
                          const double* points_cam, double* uv, int device);

/* Test/diagnostic: one residual+Jacobian evaluation at the current state; copies out the
 * accumulated normal-equation blocks in the caller's index space.  Any output may be NULL.
 *   V[36*n_cams], U[36*n_tags]  row-major 6x6 J^T J diagonal blocks (Huber-corrected, unscaled)
 *   W[36*n_obs]                 row-major 6x6 J_cam^T J_tag per observation, caller's order
 *   g_cam[6*n_cams], g_tag[6*n_tags]  J^T r                                               */
int vmm_ba_eval_blocks(vmm_ba_handle h, int robustify, double huber_a, double* cost, double* V,
                       double* U, double* W, double* g_cam, double* g_tag);

/* Test/diagnostic: solves A x = b for a dense SPD A (row-major n x n, host memory) with the
 * engine's blocked Cholesky and triangular solves; reports failure through *info != 0. */
int vmm_ba_dense_spd_solve(int device, int n, const double* A, const double* b, double* x, int* info);

/* Test/diagnostic: C = Z^T Z (n x n, row-major, lower triangle valid) for a row-major k x n Z with
 * the engine's MFMA kernel. */
int vmm_ba_dense_syrk(int device, int k, int n, const double* Z, double* C);

/* Test/diagnostic: out[i] = Plus(qt[i], delta[i]) for n poses (7 doubles each, tangent 6 each) with the engine's own
 * device function -- QuaternionParameterization::Plus on the rotation, plain addition on the translation
 * (src/TagReconstructor.cpp:665-666,692-693 attach it to every q block; tangent order: translation, rotation). */
int vmm_ba_pose_plus(int64_t n, const double* qt, const double* delta, double* out, int device);

/* Diagnostic (can the reduced-system assembly hide behind the factorisation?): ms[0] rank-k update + sum
 * alone, ms[1] factorisation + triangular solves alone, ms[2] both back to back on one stream, ms[3] both at once on
 * two streams (no data dependency between them in this measurement), ms[4] the factorisation's own duration in that
 * concurrent run, ms[5] the rank-k update + sum's; ms[6], ms[7]: total and factorisation when the rank-k update is
 * enqueued first.  ms holds 8 doubles.  Dense elimination, one GPU. */
int vmm_ba_debug_overlap(vmm_ba_handle h, int reps, double* ms);

/* Test hook, host logic only (no device needed): the launch schedule of the launch-per-column Cholesky
 * (csrc/kernels_chol.hip, chol_step_schedule) for a system of n_blk 64-column blocks whose trailing n_df block columns go
 * to the one-launch kernel (n_df < 0: the library's own choice for a chip of n_cu compute units).  launches[i] =
 * { k (-1: update-only hand-over launch), lazy0, lazy1, upd0, upd1 (panel numbers, -1: none), c0, t0, t1 }; at most
 * `cap` rows are written, *n_launches is the full count, *n_df_used the tail length.  vmm_ba_debug_chol_tile returns the
 * (block row, block column) of tile t of such a launch's update list. */
int vmm_ba_debug_chol_schedule(int n_blk, int n_df, int n_cu, int32_t* launches, int cap, int* n_launches, int* n_df_used);
int vmm_ba_debug_chol_tile(int n_blk, const int32_t* launch, int t, int* bi, int* bj);

/* Times each kernel of an LM iteration at the current state (reps launches each). */
int vmm_ba_time_kernels(vmm_ba_handle h, const vmm_ba_options* opt, int reps, vmm_ba_kernel_times* out);

#ifdef __cplusplus
}
#endif
#endif
