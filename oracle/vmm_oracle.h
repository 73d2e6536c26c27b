/*
 * vmm_oracle.h -- CPU restatement of the TagReconstructor bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (visual_marker_mapping_amd/, include/) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The reference (schmidtniko/visual_marker_mapping) delegates the numerical
 * algorithm of this path to Ceres Solver, which is neither vendored in /root/reference nor
 * installed here, and the reference ships no tests, golden vectors or sample data.  What pins
 * this file instead: (i) mpmath 50-digit known-answer tests of the residual and its tangent
 * Jacobians (tests/golden/kat_residual.json), (ii) zero-noise scenes whose optimum is the
 * ground truth, (iii) scipy.optimize.least_squares as an independent minimiser of the same
 * cost.  The trust-region policy follows Ceres 1.11..2.1 (the range the reference's
 * CMakeLists.txt:35 and src/TagReconstructor.cpp:661,726,734 admit) as recalled from upstream;
 * the reference's call sites are cited per function.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef VMM_ORACLE_H_
#define VMM_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Pose layout everywhere: 7 doubles = quaternion (w,x,y,z) then translation (x,y,z).
 * include/visual_marker_mapping/Camera.h:13-17, TagReconstructor.h:19-31.
 * Camera pose maps world->camera, tag pose maps tag->world (TagReconstructionCostFunction.h:107-122). */

typedef struct vo_problem {
    double intr[4];        /* fx, fy, cx, cy                         CameraModel.h:14-17        */
    double dist[5];        /* k1, k2, p1, p2, k3                     CameraModel.cpp:11-16      */
    int n_cams;
    double* cam_qt;        /* [7*n_cams] in/out                      TagReconstructor.cpp:692-693 */
    int n_tags;
    double* tag_qt;        /* [7*n_tags] in/out                      TagReconstructor.cpp:665-666 */
    const double* tag_wh;  /* [2*n_tags] width,height                TagReconstructor.cpp:713,718 */
    int fixed_tag;         /* dense index of the origin tag or -1    TagReconstructor.cpp:669-673 */
    int n_obs;
    const int* obs_cam;    /* [n_obs] dense camera index */
    const int* obs_tag;    /* [n_obs] dense tag index */
    const double* obs_px;  /* [8*n_obs] LL,LR,UR,UL x (u,v)          DetectionIO.cpp:45-51, README.md:216 */
    /* Point-landmark variant (OpenCVReprojectionError, TagReconstructionCostFunction.h:9-84, used by the dead
     * doBundleAdjustment_points, src/TagReconstructor.cpp:457-644): landmark_points != 0 turns every "tag" slot
     * into a PAIR of free 3-D points -- tag_qt[7*t .. 7*t+5] = (point a, point b), slot 6 unused -- and every
     * observation into the two corner observations of those points (obs_px[8*i .. 8*i+3]; the rest unused).  Two
     * points share a 6-dof block only for storage: their 3x3 blocks do not couple, every solver gives the step of
     * Ceres' 3-dof point blocks.  fixed_tag2 = a second constant block (the origin tag has two pairs), or -1. */
    int landmark_points;
    int fixed_tag2;
} vo_problem;

enum { VO_SOLVER_DENSE_NORMAL = 0, VO_SOLVER_SCHUR_ELIM_TAGS = 1, VO_SOLVER_SCHUR_ELIM_CAMS = 2,
       VO_SOLVER_SCHUR_AUTO = 3 };

/* Ceres termination_type values as printed by TagReconstructor.cpp:740. */
enum { VO_CONVERGENCE = 0, VO_NO_CONVERGENCE = 1, VO_FAILURE = 2 };

typedef struct vo_options {
    int max_num_iterations;            /* TagReconstructor.cpp:732 (400 / 1500) */
    int robustify;                     /* TagReconstructor.cpp:721 */
    double huber_a;                    /* 1.0, TagReconstructor.cpp:721 */
    double function_tolerance;         /* Ceres default 1e-6  */
    double gradient_tolerance;         /* Ceres default 1e-10 */
    double parameter_tolerance;        /* Ceres default 1e-8  */
    double initial_trust_region_radius;/* 1e4   */
    double max_trust_region_radius;    /* 1e16  */
    double min_trust_region_radius;    /* 1e-32 */
    double min_relative_decrease;      /* 1e-3  */
    double min_lm_diagonal;            /* 1e-6  */
    double max_lm_diagonal;            /* 1e32  */
    int max_num_consecutive_invalid_steps; /* 5 */
    int jacobi_scaling;                /* 1 */
    int linear_solver;                 /* VO_SOLVER_* */
    int num_threads;                   /* TagReconstructor.cpp:733 */
} vo_options;

typedef struct vo_iteration {
    int iteration;
    int step_is_valid;
    int step_is_successful;
    double cost;              /* cost at the iterate after this iteration (candidate cost if rejected) */
    double cost_change;
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;
    double model_cost_change;
} vo_iteration;

typedef struct vo_summary {
    int termination_type;
    int iterations;               /* number of vo_iteration records, i.e. Ceres' summary.iterations.size() */
    int num_successful_steps;
    int num_unsuccessful_steps;
    int num_jacobian_evals;
    int num_cost_evals;
    double initial_cost;
    double final_cost;
    double time_total_s;
    double time_eval_s;
    double time_linear_s;
    vo_iteration* trace;          /* optional caller buffer */
    int trace_capacity;
} vo_summary;

void vo_default_options(vo_options* o);

/* One corner residual, TagReconstructionCostFunction.h:101-159 with T=double. */
void vo_corner_residual(const double intr[4], const double dist[5], const double cam_qt[7],
                        const double tag_qt[7], const double corner_local[3], const double obs_uv[2],
                        double residual[2]);

/* All four corners of one tag observation: residuals r[8] (corner-major, u then v) and, when
 * non-NULL, tangent Jacobians Jc[8][6], Jt[8][6] (row-major; columns = t(3) then half-angle
 * rotation delta(3)), i.e. what AutoDiffCostFunction<...,2,3,4,3,4> (CostFunction.h:167) composed
 * with QuaternionParameterization (TagReconstructor.cpp:661) yields.  No loss applied. */
void vo_obs_eval(const double intr[4], const double dist[5], const double cam_qt[7],
                 const double tag_qt[7], const double wh[2], const double px[8], double r[8],
                 double* Jc, double* Jt);

/* Ceres HuberLoss::Evaluate(s, rho[3]) with a (TagReconstructor.cpp:721). */
void vo_huber(double a, double s, double rho[3]);

/* Ceres QuaternionParameterization::Plus on one pose: t += d[0..2]; q = exp(d[3..5]) (x) q. */
void vo_pose_plus(const double qt[7], const double delta[6], double out[7]);

/* OpenCVReprojectionError::operator() (TagReconstructionCostFunction.h:21-68) for one (camera, point, corner):
 * residual[2]; optional tangent Jacobians Jc[2][6] (camera t, then half-angle rotation) and Jp[2][3] (point). */
void vo_point_eval(const double intr[4], const double dist[5], const double cam_qt[7], const double point[3],
                   const double obs_uv[2], double residual[2], double* Jc, double* Jp);

/* Total cost 1/2 sum rho(|r|^2) over all corner blocks (Ceres Evaluator, cost only). */
double vo_cost(const vo_problem* p, const vo_options* o);

/* doBundleAdjustment, TagReconstructor.cpp:646-743.  Mutates p->cam_qt / p->tag_qt in place. */
int vo_solve(vo_problem* p, const vo_options* o, vo_summary* s);

/* Reprojection statistics, TagReconstructor.cpp:340-455.  per_cam_mean[n_cams] (-1 where a camera
 * has no observation, :379-383), per_tag_mean[n_tags] (NaN where a tag has none), *avg grand mean
 * (:416-426), per_corner[8*n_obs] signed pixel errors (:447-451).  Any output may be NULL. */
void vo_reprojection_stats(const vo_problem* p, double* per_cam_mean, double* per_tag_mean,
                           double* avg, double* per_corner);

/* CameraModel::projectPoint, CameraModel.cpp:6-26. */
/* Covariance of the tag translations: cov[9*t..] = 3x3 block of (J^T J)^-1 (TagReconstructor.cpp:744-783).
 * Returns nonzero if J^T J is not positive definite. */
int vo_tag_translation_covariance(const vo_problem* p, const vo_options* o, double* cov);

void vo_project_point(const double intr[4], const double dist[5], const double pc[3], double uv[2]);

#ifdef __cplusplus
}
#endif
#endif
