/*
 * vmm_oracle.c -- CPU restatement of TagReconstructor::doBundleAdjustment and its residual.
 *
 * TEST INFRASTRUCTURE ONLY (see vmm_oracle.h).  PARITY UNPINNED against Ceres (see vmm_oracle.h).
 * Plain C11 + optional OpenMP.  Built with -ffp-contract=off so every operation rounds once, in the
 * order written, like the reference's templated functor instantiated with T=double.
 *
 * All file:line citations are relative to /root/reference.
 */
#include "vmm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void vo_default_options(vo_options* o)
{
    /* TagReconstructor.cpp:725-735 sets only max_num_iterations, num_threads, progress=false and the
     * ordering; everything else is the Ceres Solver::Options default. */
    o->max_num_iterations = 400;
    o->robustify = 1;
    o->huber_a = 1.0;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
    o->linear_solver = VO_SOLVER_SCHUR_AUTO;
    o->num_threads = 1;
}

/* ------------------------------------------------------------------------------------------------
 * Residual
 * ---------------------------------------------------------------------------------------------- */

/* ceres::UnitQuaternionRotatePoint (ceres/rotation.h, Ceres 1.x form), called through
 * ceres::QuaternionRotatePoint at TagReconstructionCostFunction.h:109,118. */
static void unit_quat_rotate(const double q[4], const double pt[3], double out[3])
{
    const double t2 = q[0] * q[1];
    const double t3 = q[0] * q[2];
    const double t4 = q[0] * q[3];
    const double t5 = -q[1] * q[1];
    const double t6 = q[1] * q[2];
    const double t7 = q[1] * q[3];
    const double t8 = -q[2] * q[2];
    const double t9 = q[2] * q[3];
    const double t1 = -q[3] * q[3];
    out[0] = 2.0 * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
    out[1] = 2.0 * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
    out[2] = 2.0 * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}

/* ceres::QuaternionRotatePoint: normalise, then rotate. */
static void quat_rotate(const double q[4], const double pt[3], double out[3])
{
    const double scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double u[4] = { scale * q[0], scale * q[1], scale * q[2], scale * q[3] };
    unit_quat_rotate(u, pt, out);
}

/* The part of the functor after the two rigid transforms: TagReconstructionCostFunction.h:125-152.
 * NOT the arithmetic of CameraModel::projectPoint (see project_camera_model below). */
static void project_distort(const double intr[4], const double dist[5], const double pc[3],
                            double uv[2])
{
    const double xp = pc[0] / pc[2];
    const double yp = pc[1] / pc[2];
    const double r2 = xp * xp + yp * yp;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    const double xd = xp * (1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2.0 * p1 * xp * yp
        + p2 * (r2 + 2.0 * xp * xp);
    const double yd = yp * (1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2.0 * p2 * xp * yp
        + p1 * (r2 + 2.0 * yp * yp);
    uv[0] = intr[0] * xd + intr[2];
    uv[1] = intr[1] * yd + intr[3];
}

/* CameraModel::projectPoint, CameraModel.cpp:6-26, statement by statement.  pt is a Vector3d divided by its
 * z (:9); r2 from the undistorted x, y (:18); pt.x() is OVERWRITTEN with the distorted value (:20-21) before
 * pt.y() is computed (:22-23), so 2*p2*pt.x()*pt.y() uses the distorted x there (r2 and the remaining pt.y()
 * factors are still undistorted); then (getK() * pt).head(2) (:25) = (fx*x + 0*y + cx*1, 0*x + fy*y + cy*1). */
static void project_camera_model(const double intr[4], const double dist[5], const double pc[3],
                                 double uv[2])
{
    double px = pc[0] / pc[2];
    double py = pc[1] / pc[2];
    const double pz = pc[2] / pc[2];
    const double k1 = dist[0], k2 = dist[1], k3 = dist[4], p1 = dist[2], p2 = dist[3];
    const double r2 = px * px + py * py;
    px = px * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2 * p1 * px * py + p2 * (r2 + 2 * px * px);
    py = py * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2 * p2 * px * py + p1 * (r2 + 2 * py * py);
    uv[0] = (intr[0] * px + 0.0 * py) + intr[2] * pz;
    uv[1] = (0.0 * px + intr[1] * py) + intr[3] * pz;
}

void vo_project_point(const double intr[4], const double dist[5], const double pc[3], double uv[2])
{
    project_camera_model(intr, dist, pc, uv);
}

void vo_corner_residual(const double intr[4], const double dist[5], const double cam_qt[7],
                        const double tag_qt[7], const double corner_local[3], const double obs_uv[2],
                        double residual[2])
{
    double pw[3], pc[3], uv[2];
    quat_rotate(tag_qt, corner_local, pw);      /* CostFunction.h:109 */
    pw[0] += tag_qt[4];                         /* :112-114 */
    pw[1] += tag_qt[5];
    pw[2] += tag_qt[6];
    quat_rotate(cam_qt, pw, pc);                /* :118 */
    pc[0] += cam_qt[4];                         /* :120-122 */
    pc[1] += cam_qt[5];
    pc[2] += cam_qt[6];
    project_distort(intr, dist, pc, uv);        /* :125-152 */
    residual[0] = uv[0] - obs_uv[0];            /* :155-156 */
    residual[1] = uv[1] - obs_uv[1];
}

/* Local corner quad, TagReconstructor.h:44-52: LL, LR, UR, UL. */
static void local_corner(const double wh[2], int i, double c[3])
{
    static const double sx[4] = { -1.0, 1.0, 1.0, -1.0 };
    static const double sy[4] = { -1.0, -1.0, 1.0, 1.0 };
    c[0] = sx[i] * wh[0] / 2.0;
    c[1] = sy[i] * wh[1] / 2.0;
    c[2] = 0.0;
}

/* Rotation matrix of the normalised quaternion (row-major 3x3). */
static void quat_to_R(const double q[4], double R[9])
{
    const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
    R[0] = 1.0 - 2.0 * (y * y + z * z);
    R[1] = 2.0 * (x * y - w * z);
    R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);
    R[4] = 1.0 - 2.0 * (x * x + z * z);
    R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);
    R[7] = 2.0 * (y * z + w * x);
    R[8] = 1.0 - 2.0 * (x * x + y * y);
}

void vo_obs_eval(const double intr[4], const double dist[5], const double cam_qt[7],
                 const double tag_qt[7], const double wh[2], const double px[8], double r[8],
                 double* Jc, double* Jt)
{
    double Rc[9], Rt[9];
    if (Jc || Jt) {
        quat_to_R(cam_qt, Rc);
        quat_to_R(tag_qt, Rt);
    }
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    for (int i = 0; i < 4; ++i) {
        double cl[3];
        local_corner(wh, i, cl);
        /* residual: the functor itself */
        vo_corner_residual(intr, dist, cam_qt, tag_qt, cl, px + 2 * i, r + 2 * i);
        if (!Jc && !Jt)
            continue;
        /* Analytic tangent Jacobians.  a = R_t p_l, b = R_c (a + t_t). */
        double a[3], pw[3], b[3], pc[3];
        for (int k = 0; k < 3; ++k)
            a[k] = Rt[3 * k + 0] * cl[0] + Rt[3 * k + 1] * cl[1] + Rt[3 * k + 2] * cl[2];
        for (int k = 0; k < 3; ++k)
            pw[k] = a[k] + tag_qt[4 + k];
        for (int k = 0; k < 3; ++k)
            b[k] = Rc[3 * k + 0] * pw[0] + Rc[3 * k + 1] * pw[1] + Rc[3 * k + 2] * pw[2];
        for (int k = 0; k < 3; ++k)
            pc[k] = b[k] + cam_qt[4 + k];
        const double iz = 1.0 / pc[2];
        const double x = pc[0] * iz, y = pc[1] * iz;
        const double r2 = x * x + y * y;
        const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
        const double dr = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
        /* D = d(xd,yd)/d(x,y) */
        const double D00 = rad + 2.0 * x * x * dr + 2.0 * p1 * y + 6.0 * p2 * x;
        const double D01 = 2.0 * x * y * dr + 2.0 * p1 * x + 2.0 * p2 * y;
        const double D10 = D01;
        const double D11 = rad + 2.0 * y * y * dr + 2.0 * p2 * x + 6.0 * p1 * y;
        /* G = diag(fx,fy) * D * Pi,  Pi = [iz 0 -x iz; 0 iz -y iz] */
        double G[2][3];
        G[0][0] = intr[0] * D00 * iz;
        G[0][1] = intr[0] * D01 * iz;
        G[0][2] = -intr[0] * (D00 * x + D01 * y) * iz;
        G[1][0] = intr[1] * D10 * iz;
        G[1][1] = intr[1] * D11 * iz;
        G[1][2] = -intr[1] * (D10 * x + D11 * y) * iz;
        for (int row = 0; row < 2; ++row) {
            const double* g = G[row];
            /* H = G R_c (derivative w.r.t. a world-frame displacement) */
            double h[3];
            for (int k = 0; k < 3; ++k)
                h[k] = g[0] * Rc[0 + k] + g[1] * Rc[3 + k] + g[2] * Rc[6 + k];
            if (Jc) {
                double* o = Jc + 6 * (2 * i + row);
                o[0] = g[0];
                o[1] = g[1];
                o[2] = g[2];
                /* -2 g^T [b]x = 2 (b x g) */
                o[3] = 2.0 * (b[1] * g[2] - b[2] * g[1]);
                o[4] = 2.0 * (b[2] * g[0] - b[0] * g[2]);
                o[5] = 2.0 * (b[0] * g[1] - b[1] * g[0]);
            }
            if (Jt) {
                double* o = Jt + 6 * (2 * i + row);
                o[0] = h[0];
                o[1] = h[1];
                o[2] = h[2];
                o[3] = 2.0 * (a[1] * h[2] - a[2] * h[1]);
                o[4] = 2.0 * (a[2] * h[0] - a[0] * h[2]);
                o[5] = 2.0 * (a[0] * h[1] - a[1] * h[0]);
            }
        }
    }
}

/* OpenCVReprojectionError (TagReconstructionCostFunction.h:21-68): ceres::UnitQuaternionRotatePoint -- NO
 * normalisation of camera_q (:27) -- then the same projection as the tag functor (:38-63).  The Jacobians are what
 * AutoDiffCostFunction<..., 2, 3, 4, 3> (:74-75) followed by QuaternionParameterization (src/TagReconstructor.cpp:529)
 * give: d/d(point) = A R_u(q); d/d(delta) = A [dR_u(q)p/dq] G(q) with G the 4x3 Jacobian of Plus at delta = 0
 * (SURVEY.md Appendix A.2); R_u(q) = I + 2 M(q) is the polynomial UnitQuaternionRotatePoint evaluates, so for a
 * quaternion that is not exactly unit this is NOT 2 (R p) x g. */
void vo_point_eval(const double intr[4], const double dist[5], const double cam_qt[7], const double point[3],
                   const double obs_uv[2], double residual[2], double* Jc, double* Jp)
{
    double pc[3], uv[2];
    unit_quat_rotate(cam_qt, point, pc);        /* :27 */
    pc[0] += cam_qt[4];                         /* :29-31 */
    pc[1] += cam_qt[5];
    pc[2] += cam_qt[6];
    project_distort(intr, dist, pc, uv);        /* :34-63 */
    residual[0] = uv[0] - obs_uv[0];            /* :66-67 */
    residual[1] = uv[1] - obs_uv[1];
    if (!Jc && !Jp)
        return;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4];
    const double iz = 1.0 / pc[2];
    const double x = pc[0] * iz, y = pc[1] * iz;
    const double r2 = x * x + y * y;
    const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double dr = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
    const double D00 = rad + 2.0 * x * x * dr + 2.0 * p1 * y + 6.0 * p2 * x;
    const double D01 = 2.0 * x * y * dr + 2.0 * p1 * x + 2.0 * p2 * y;
    const double D11 = rad + 2.0 * y * y * dr + 2.0 * p2 * x + 6.0 * p1 * y;
    double G[2][3];
    G[0][0] = intr[0] * D00 * iz;
    G[0][1] = intr[0] * D01 * iz;
    G[0][2] = -intr[0] * (D00 * x + D01 * y) * iz;
    G[1][0] = intr[1] * D01 * iz;
    G[1][1] = intr[1] * D11 * iz;
    G[1][2] = -intr[1] * (D01 * x + D11 * y) * iz;
    const double qw = cam_qt[0], qx = cam_qt[1], qy = cam_qt[2], qz = cam_qt[3];
    const double p0 = point[0], p1w = point[1], p2w = point[2];
    /* R_u(q) = I + 2 M(q) */
    const double Ru[9] = { 1.0 - 2.0 * (qy * qy + qz * qz), 2.0 * (qx * qy - qw * qz), 2.0 * (qx * qz + qw * qy),
                           2.0 * (qx * qy + qw * qz), 1.0 - 2.0 * (qx * qx + qz * qz), 2.0 * (qy * qz - qw * qx),
                           2.0 * (qx * qz - qw * qy), 2.0 * (qy * qz + qw * qx), 1.0 - 2.0 * (qx * qx + qy * qy) };
    /* columns of d(R_u p)/d(w,x,y,z) */
    const double Dq[3][4] = {
        { 2.0 * (qy * p2w - qz * p1w), 2.0 * (qy * p1w + qz * p2w), 2.0 * (-2.0 * qy * p0 + qx * p1w + qw * p2w),
          2.0 * (-2.0 * qz * p0 - qw * p1w + qx * p2w) },
        { 2.0 * (qz * p0 - qx * p2w), 2.0 * (qy * p0 - 2.0 * qx * p1w - qw * p2w), 2.0 * (qx * p0 + qz * p2w),
          2.0 * (qw * p0 - 2.0 * qz * p1w + qy * p2w) },
        { 2.0 * (qx * p1w - qy * p0), 2.0 * (qz * p0 + qw * p1w - 2.0 * qx * p2w), 2.0 * (-qw * p0 + qz * p1w - 2.0 * qy * p2w),
          2.0 * (qx * p0 + qy * p1w) } };
    /* Plus Jacobian at delta = 0 (rows w,x,y,z) */
    const double Gq[4][3] = { { -qx, -qy, -qz }, { qw, qz, -qy }, { -qz, qw, qx }, { qy, -qx, qw } };
    for (int row = 0; row < 2; ++row) {
        const double* g = G[row];
        if (Jc) {
            double* o = Jc + 6 * row;
            o[0] = g[0];
            o[1] = g[1];
            o[2] = g[2];
            for (int k = 0; k < 3; ++k) {
                double v = 0.0;
                for (int a = 0; a < 3; ++a) {
                    double dk = 0.0;
                    for (int b = 0; b < 4; ++b)
                        dk += Dq[a][b] * Gq[b][k];
                    v += g[a] * dk;
                }
                o[3 + k] = v;
            }
        }
        if (Jp)
            for (int k = 0; k < 3; ++k)
                Jp[3 * row + k] = g[0] * Ru[0 + k] + g[1] * Ru[3 + k] + g[2] * Ru[6 + k];
    }
}

/* One "observation" of the point variant: the two corners of a point pair (see vo_problem.landmark_points). */
static void obs_eval_point_pair(const double intr[4], const double dist[5], const double cam_qt[7],
                                const double pair[6], const double px[4], double r[8], double* Jc, double* Jt)
{
    memset(r, 0, 8 * sizeof(double));
    if (Jc)
        memset(Jc, 0, 48 * sizeof(double));
    if (Jt)
        memset(Jt, 0, 48 * sizeof(double));
    for (int i = 0; i < 2; ++i) {
        double jc[12], jp[6];
        vo_point_eval(intr, dist, cam_qt, pair + 3 * i, px + 2 * i, r + 2 * i, Jc ? jc : NULL, Jt ? jp : NULL);
        for (int row = 0; row < 2; ++row) {
            if (Jc)
                memcpy(Jc + 6 * (2 * i + row), jc + 6 * row, 6 * sizeof(double));
            if (Jt)
                memcpy(Jt + 6 * (2 * i + row) + 3 * i, jp + 3 * row, 3 * sizeof(double));
        }
    }
}

/* the residual block(s) of observation i in either landmark model */
static void obs_eval(const vo_problem* p, const double* cam_qt, const double* tag_qt, int i, double r[8],
                     double* Jc, double* Jt)
{
    const int c = p->obs_cam[i], t = p->obs_tag[i];
    if (p->landmark_points)
        obs_eval_point_pair(p->intr, p->dist, cam_qt + 7 * c, tag_qt + 7 * t, p->obs_px + 8 * i, r, Jc, Jt);
    else
        vo_obs_eval(p->intr, p->dist, cam_qt + 7 * c, tag_qt + 7 * t, p->tag_wh + 2 * t, p->obs_px + 8 * i, r, Jc, Jt);
}

/* Plus of a landmark block: quaternion + translation (tag poses) or plain addition (a pair of points) */
static void landmark_plus(int landmark_points, const double x[7], const double d[6], double out[7])
{
    if (!landmark_points) {
        vo_pose_plus(x, d, out);
        return;
    }
    for (int k = 0; k < 6; ++k)
        out[k] = x[k] + d[k];
    out[6] = x[6];
}

void vo_huber(double a, double s, double rho[3])
{
    /* ceres::HuberLoss::Evaluate, constructed with a=1.0 at TagReconstructor.cpp:721. */
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r;
        if (rho[1] < DBL_MIN)
            rho[1] = DBL_MIN;
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s;
        rho[1] = 1.0;
        rho[2] = 0.0;
    }
}

void vo_pose_plus(const double qt[7], const double d[6], double out[7])
{
    /* translation block: plain addition (no parameterization, TagReconstructor.cpp:666,693) */
    out[4] = qt[4] + d[0];
    out[5] = qt[5] + d[1];
    out[6] = qt[6] + d[2];
    /* ceres::QuaternionParameterization::Plus (TagReconstructor.cpp:661,665,692) */
    const double nd = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (nd > 0.0) {
        const double s = sin(nd) / nd;
        const double z[4] = { cos(nd), s * d[3], s * d[4], s * d[5] };
        const double* w = qt;
        out[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
        out[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
        out[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
        out[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
    } else {
        out[0] = qt[0];
        out[1] = qt[1];
        out[2] = qt[2];
        out[3] = qt[3];
    }
}

/* ------------------------------------------------------------------------------------------------
 * Cost-only evaluation (Ceres Evaluator with residuals/jacobians == NULL)
 * ---------------------------------------------------------------------------------------------- */

static double obs_cost(const vo_options* o, const double r[8])
{
    double c = 0.0;
    for (int k = 0; k < 4; ++k) {
        /* ResidualBlock::Evaluate: cost = 1/2 rho(|r|^2) per corner block */
        const double s = r[2 * k] * r[2 * k] + r[2 * k + 1] * r[2 * k + 1];
        if (o->robustify) {
            double rho[3];
            vo_huber(o->huber_a, s, rho);
            c += 0.5 * rho[0];
        } else {
            c += 0.5 * s;
        }
    }
    return c;
}

static double cost_at(const vo_problem* p, const vo_options* o, const double* cam_qt,
                      const double* tag_qt)
{
    /* Deterministic for any thread count: per-observation costs, summed serially in order. */
    const int n_obs = p->n_obs;
    double* part = (double*)calloc((size_t)(n_obs > 0 ? n_obs : 1), sizeof(double));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_obs; ++i) {
        double r[8];
        obs_eval(p, cam_qt, tag_qt, i, r, NULL, NULL);
        part[i] = obs_cost(o, r);
    }
    double cost = 0.0;
    for (int i = 0; i < n_obs; ++i)
        cost += part[i];
    free(part);
    return cost;
}

double vo_cost(const vo_problem* p, const vo_options* o)
{
    return cost_at(p, o, p->cam_qt, p->tag_qt);
}

/* ------------------------------------------------------------------------------------------------
 * Small dense helpers
 * ---------------------------------------------------------------------------------------------- */

/* In-place lower Cholesky of a row-major n x n SPD matrix (only the lower triangle is read and
 * written).  Returns 0 on success, 1 when a pivot is not positive / not finite. */
static int chol_lower(double* A, int n, int lda)
{
    const int nb = 48;
    for (int k0 = 0; k0 < n; k0 += nb) {
        const int kb = (n - k0 < nb) ? n - k0 : nb;
        /* diagonal block */
        for (int j = k0; j < k0 + kb; ++j) {
            double d = A[(size_t)j * lda + j];
            for (int k = k0; k < j; ++k)
                d -= A[(size_t)j * lda + k] * A[(size_t)j * lda + k];
            if (!(d > 0.0) || !isfinite(d))
                return 1;
            d = sqrt(d);
            A[(size_t)j * lda + j] = d;
            for (int i = j + 1; i < k0 + kb; ++i) {
                double s = A[(size_t)i * lda + j];
                for (int k = k0; k < j; ++k)
                    s -= A[(size_t)i * lda + k] * A[(size_t)j * lda + k];
                A[(size_t)i * lda + j] = s / d;
            }
        }
        const int r0 = k0 + kb;
        if (r0 >= n)
            break;
            /* panel: rows below the diagonal block */
#pragma omp parallel for schedule(static)
        for (int i = r0; i < n; ++i) {
            double* ai = A + (size_t)i * lda;
            for (int j = k0; j < k0 + kb; ++j) {
                const double* aj = A + (size_t)j * lda;
                double s = ai[j];
                for (int k = k0; k < j; ++k)
                    s -= ai[k] * aj[k];
                ai[j] = s / aj[j];
            }
        }
        /* trailing update: A[i][j] -= L[i][k0:k0+kb] . L[j][k0:k0+kb] */
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = r0; i < n; ++i) {
            double* ai = A + (size_t)i * lda;
            for (int j = r0; j <= i; ++j) {
                const double* aj = A + (size_t)j * lda;
                double s = 0.0;
                for (int k = k0; k < k0 + kb; ++k)
                    s += ai[k] * aj[k];
                ai[j] -= s;
            }
        }
    }
    return 0;
}

/* Solve L L^T x = b in place given the lower factor. */
static void chol_solve(const double* L, int n, int lda, double* x)
{
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        const double* li = L + (size_t)i * lda;
        for (int k = 0; k < i; ++k)
            s -= li[k] * x[k];
        x[i] = s / li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k)
            s -= L[(size_t)k * lda + i] * x[k];
        x[i] = s / L[(size_t)i * lda + i];
    }
}

/* ------------------------------------------------------------------------------------------------
 * LM solve
 * ---------------------------------------------------------------------------------------------- */

typedef struct work {
    int n_c, n_t, n_obs, n_pose, n_tan;
    int landmark_points;
    double* J;       /* per observation: Jc[48] | Jt[48] (corrected; scaled after scaling step) */
    double* r;       /* per observation 8 corrected residuals */
    double* g;       /* tangent gradient (unscaled), 6 per pose; cameras first then tags */
    double* scale;   /* Jacobi scaling per tangent column */
    double* diag;    /* LM diagonal (clamped squared column norms of the scaled Jacobian) */
    double* step;    /* trust-region step in scaled coordinates */
    double* delta;   /* step in unscaled tangent coordinates */
    int* active;     /* per pose: takes part in the reduced program */
    int *off, *lst;  /* observations of every pose (cameras then tags), in observation order: a pose's sums are
                      * formed by ONE thread in that order, so every result is independent of the thread count */
} work;

static void build_obs_lists(const vo_problem* p, work* w)
{
    const int n_c = p->n_cams, n_obs = p->n_obs;
    w->off = (int*)calloc((size_t)w->n_pose + 2, sizeof(int));
    w->lst = (int*)malloc((size_t)2 * (n_obs > 0 ? n_obs : 1) * sizeof(int));
    for (int i = 0; i < n_obs; ++i) {
        w->off[p->obs_cam[i] + 1]++;
        w->off[n_c + p->obs_tag[i] + 1]++;
    }
    for (int q = 0; q < w->n_pose; ++q)
        w->off[q + 1] += w->off[q];
    int* pos = (int*)malloc(((size_t)w->n_pose + 1) * sizeof(int));
    memcpy(pos, w->off, (size_t)w->n_pose * sizeof(int));
    for (int i = 0; i < n_obs; ++i) {
        w->lst[pos[p->obs_cam[i]]++] = i;
        w->lst[pos[n_c + p->obs_tag[i]]++] = i;
    }
    free(pos);
}

static int eval_full(const vo_problem* p, const vo_options* o, work* w, const double* cam_qt,
                     const double* tag_qt, double* cost_out)
{
    const int n_obs = p->n_obs;
    double* part = (double*)calloc((size_t)(n_obs > 0 ? n_obs : 1), sizeof(double));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_obs; ++i) {
        const int t = p->obs_tag[i];
        double* Jc = w->J + (size_t)96 * i;
        double* Jt = Jc + 48;
        double* r = w->r + (size_t)8 * i;
        obs_eval(p, cam_qt, tag_qt, i, r, Jc, Jt);
        if (!w->active[p->n_cams + t])
            memset(Jt, 0, 48 * sizeof(double));
        part[i] = obs_cost(o, r);
        if (o->robustify) {
            for (int k = 0; k < 4; ++k) {
                /* Corrector: rho'' <= 0 for Huber, so rows are scaled by sqrt(rho') only; the
                 * Jacobian is corrected with the uncorrected residual norm, then the residual. */
                const double s = r[2 * k] * r[2 * k] + r[2 * k + 1] * r[2 * k + 1];
                double rho[3];
                vo_huber(o->huber_a, s, rho);
                const double sr = sqrt(rho[1]);
                for (int q = 0; q < 12; ++q) {
                    Jc[12 * k + q] *= sr;
                    Jt[12 * k + q] *= sr;
                }
                r[2 * k] *= sr;
                r[2 * k + 1] *= sr;
            }
        }
    }
    double cost = 0.0;
    for (int i = 0; i < n_obs; ++i)
        cost += part[i];
    free(part);
    *cost_out = cost;
    /* gradient g = J^T r: per pose, in observation order */
    memset(w->g, 0, (size_t)w->n_tan * sizeof(double));
#pragma omp parallel for schedule(dynamic, 8)
    for (int q = 0; q < w->n_pose; ++q) {
        double* gq = w->g + 6 * q;
        const int is_tag = q >= p->n_cams;
        for (int k2 = w->off[q]; k2 < w->off[q + 1]; ++k2) {
            const int i = w->lst[k2];
            const double* Jq = w->J + (size_t)96 * i + (is_tag ? 48 : 0);
            const double* r = w->r + (size_t)8 * i;
            for (int row = 0; row < 8; ++row)
                for (int k = 0; k < 6; ++k)
                    gq[k] += Jq[6 * row + k] * r[row];
        }
    }
    return !isfinite(cost);
}

static void col_sq_norms(const vo_problem* p, const work* w, double* out)
{
    memset(out, 0, (size_t)w->n_tan * sizeof(double));
#pragma omp parallel for schedule(dynamic, 8)
    for (int q = 0; q < w->n_pose; ++q) {
        double* oq = out + 6 * q;
        const int is_tag = q >= p->n_cams;
        for (int k2 = w->off[q]; k2 < w->off[q + 1]; ++k2) {
            const double* Jq = w->J + (size_t)96 * w->lst[k2] + (is_tag ? 48 : 0);
            for (int row = 0; row < 8; ++row)
                for (int k = 0; k < 6; ++k)
                    oq[k] += Jq[6 * row + k] * Jq[6 * row + k];
        }
    }
}

static void scale_columns(const vo_problem* p, work* w)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < p->n_obs; ++i) {
        const int c = p->obs_cam[i], t = p->obs_tag[i];
        double* Jc = w->J + (size_t)96 * i;
        double* Jt = Jc + 48;
        const double* sc = w->scale + 6 * c;
        const double* st = w->scale + 6 * (p->n_cams + t);
        for (int row = 0; row < 8; ++row)
            for (int k = 0; k < 6; ++k) {
                Jc[6 * row + k] *= sc[k];
                Jt[6 * row + k] *= st[k];
            }
    }
}

/* Solve (J^T J + D^2) y = J^T r densely in the full tangent space.  Returns nonzero on failure. */
static int solve_dense_normal(const vo_problem* p, const work* w, const double* D2, double* y)
{
    const int n = w->n_tan;
    double* H = (double*)calloc((size_t)n * n, sizeof(double));
    double* b = y;
    memset(b, 0, (size_t)n * sizeof(double));
    for (int i = 0; i < p->n_obs; ++i) {
        const int c = p->obs_cam[i], t = p->obs_tag[i];
        const double* Jc = w->J + (size_t)96 * i;
        const double* Jt = Jc + 48;
        const double* r = w->r + (size_t)8 * i;
        const int ic = 6 * c, it = 6 * (p->n_cams + t);
        for (int row = 0; row < 8; ++row) {
            const double* jc = Jc + 6 * row;
            const double* jt = Jt + 6 * row;
            for (int a = 0; a < 6; ++a) {
                b[ic + a] += jc[a] * r[row];
                b[it + a] += jt[a] * r[row];
                for (int q = 0; q <= a; ++q) {
                    H[(size_t)(ic + a) * n + ic + q] += jc[a] * jc[q];
                    H[(size_t)(it + a) * n + it + q] += jt[a] * jt[q];
                }
                for (int q = 0; q < 6; ++q)
                    H[(size_t)(it + a) * n + ic + q] += jt[a] * jc[q]; /* tags after cameras: lower */
            }
        }
    }
    for (int i = 0; i < n; ++i)
        H[(size_t)i * n + i] += D2[i];
    int fail = chol_lower(H, n, n);
    if (!fail)
        chol_solve(H, n, n, b);
    free(H);
    return fail;
}

/* 6x6 helpers (row-major) */
static int chol6(double* A)
{
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k)
            d -= A[6 * j + k] * A[6 * j + k];
        if (!(d > 0.0) || !isfinite(d))
            return 1;
        d = sqrt(d);
        A[6 * j + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; ++k)
                s -= A[6 * i + k] * A[6 * j + k];
            A[6 * i + j] = s / d;
        }
    }
    return 0;
}

/* X := L^{-1} X for a 6 x m row-major X. */
static void fwd6(const double* L, double* X, int m)
{
    for (int i = 0; i < 6; ++i) {
        for (int k = 0; k < i; ++k)
            for (int c = 0; c < m; ++c)
                X[i * m + c] -= L[6 * i + k] * X[k * m + c];
        for (int c = 0; c < m; ++c)
            X[i * m + c] /= L[6 * i + i];
    }
}

/* x := L^{-T} x for a 6-vector. */
static void bwd6(const double* L, double* x)
{
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < 6; ++k)
            s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
}

/* Block elimination of one pose family (exact, SURVEY Appendix A.4 "Schur equivalence").
 * elim_tags != 0: e-blocks are the tags (the reference's ordering group 0,
 * TagReconstructor.cpp:675-676), f-blocks the cameras; otherwise the roles swap. */
static int solve_schur(const vo_problem* p, const work* w, const double* D2, double* y, int elim_tags)
{
    const int n_c = p->n_cams, n_t = p->n_tags, n_obs = p->n_obs;
    const int n_e = elim_tags ? n_t : n_c;
    const int n_f = elim_tags ? n_c : n_t;
    const int off_e = elim_tags ? 6 * n_c : 0; /* offset of family in the tangent vector */
    const int off_f = elim_tags ? 0 : 6 * n_c;
    const int nf6 = 6 * n_f;
    int fail = 0;

    /* observation lists per e-block, in observation order */
    int* cnt = (int*)calloc((size_t)n_e + 1, sizeof(int));
    for (int i = 0; i < n_obs; ++i)
        cnt[(elim_tags ? p->obs_tag[i] : p->obs_cam[i]) + 1]++;
    for (int e = 0; e < n_e; ++e)
        cnt[e + 1] += cnt[e];
    int* lst = (int*)malloc((size_t)(n_obs > 0 ? n_obs : 1) * sizeof(int));
    int* pos = (int*)malloc((size_t)(n_e > 0 ? n_e : 1) * sizeof(int));
    memcpy(pos, cnt, (size_t)n_e * sizeof(int));
    for (int i = 0; i < n_obs; ++i)
        lst[pos[elim_tags ? p->obs_tag[i] : p->obs_cam[i]]++] = i;

    double* Me = (double*)calloc((size_t)36 * (n_e > 0 ? n_e : 1), sizeof(double)); /* -> L_e */
    double* ze = (double*)calloc((size_t)6 * (n_e > 0 ? n_e : 1), sizeof(double));  /* L_e^{-1} g_e */
    double* Z = (double*)malloc((size_t)36 * (n_obs > 0 ? n_obs : 1) * sizeof(double)); /* L_e^{-1} W_ef */
    double* S = (double*)calloc((size_t)nf6 * nf6, sizeof(double));
    double* b = (double*)calloc((size_t)nf6, sizeof(double));

    /* F diagonal blocks and rhs: per kept pose, in observation order */
#pragma omp parallel for schedule(dynamic, 4)
    for (int f = 0; f < n_f; ++f) {
        const int q0 = elim_tags ? f : n_c + f;
        for (int k2 = w->off[q0]; k2 < w->off[q0 + 1]; ++k2) {
            const int i = w->lst[k2];
            const double* Jf = w->J + (size_t)96 * i + (elim_tags ? 0 : 48);
            const double* r = w->r + (size_t)8 * i;
            for (int row = 0; row < 8; ++row)
                for (int a = 0; a < 6; ++a) {
                    b[6 * f + a] += Jf[6 * row + a] * r[row];
                    for (int q = 0; q <= a; ++q)
                        S[(size_t)(6 * f + a) * nf6 + 6 * f + q] += Jf[6 * row + a] * Jf[6 * row + q];
                }
        }
    }
    for (int i = 0; i < nf6; ++i)
        S[(size_t)i * nf6 + i] += D2[off_f + i];
    /* E blocks: M_e, factor, Z_ef = L_e^{-1} W_ef, z_e = L_e^{-1} g_e */
#pragma omp parallel for schedule(dynamic, 4) reduction(| : fail)
    for (int e = 0; e < n_e; ++e) {
        double* M = Me + 36 * e;
        double* ge = ze + 6 * e;
        for (int k = cnt[e]; k < cnt[e + 1]; ++k) {
            const int i = lst[k];
            const double* Je = w->J + (size_t)96 * i + (elim_tags ? 48 : 0);
            const double* Jf = w->J + (size_t)96 * i + (elim_tags ? 0 : 48);
            const double* r = w->r + (size_t)8 * i;
            double* W = Z + (size_t)36 * i;
            memset(W, 0, 36 * sizeof(double));
            for (int row = 0; row < 8; ++row)
                for (int a = 0; a < 6; ++a) {
                    ge[a] += Je[6 * row + a] * r[row];
                    for (int q = 0; q < 6; ++q) {
                        M[6 * a + q] += Je[6 * row + a] * Je[6 * row + q];
                        W[6 * a + q] += Je[6 * row + a] * Jf[6 * row + q];
                    }
                }
        }
        for (int a = 0; a < 6; ++a)
            M[6 * a + a] += D2[off_e + 6 * e + a];
        if (chol6(M)) {
            fail |= 1;
            continue;
        }
        fwd6(M, ge, 1);
        for (int k = cnt[e]; k < cnt[e + 1]; ++k)
            fwd6(M, Z + (size_t)36 * lst[k], 6);
    }
    const double t_dbg0 = omp_get_wtime();
    if (!fail) {
        /* S -= Z^T Z, b -= Z^T z (lower triangle).  Row block f of S is owned by one thread and its
         * contributions are added in the order of f's observation list: deterministic for any
         * thread count. */
        /* obs_of[e][f]: the observation of the pair, or -1 (a pair observed twice keeps the list form below) */
        int* obs_of = NULL;
        int dup = 0;
        if ((size_t)n_e * (size_t)n_f <= ((size_t)1 << 27)) {
            obs_of = (int*)malloc((size_t)n_e * n_f * sizeof(int));
            for (size_t q = 0; q < (size_t)n_e * n_f; ++q)
                obs_of[q] = -1;
            for (int i = 0; i < n_obs && !dup; ++i) {
                const int e = elim_tags ? p->obs_tag[i] : p->obs_cam[i];
                const int f = elim_tags ? p->obs_cam[i] : p->obs_tag[i];
                if (obs_of[(size_t)e * n_f + f] >= 0)
                    dup = 1;
                obs_of[(size_t)e * n_f + f] = i;
            }
        }
        if (obs_of && !dup) {
            /* Blocks of FB kept poses own their block rows of S; for every eliminated pose e the blocks Z_e,f of the
             * row are read once per block of rows instead of once per row (the product is memory-bound: Z is 29 MB at
             * 500 x 200).  Contributions to a block of S are added in the order of e: deterministic for any thread
             * count. */
            enum { FB = 8 };
            const int n_fb = (n_f + FB - 1) / FB;
#pragma omp parallel for schedule(dynamic, 1)
            for (int blk = n_fb - 1; blk >= 0; --blk) {   /* the largest blocks first */
                const int f0 = blk * FB, f1 = (f0 + FB < n_f) ? f0 + FB : n_f;
                for (int e = 0; e < n_e; ++e) {
                    const int* row = obs_of + (size_t)e * n_f;
                    const double* zz = ze + 6 * e;
                    for (int fa = f0; fa < f1; ++fa) {
                        const int ia = row[fa];
                        if (ia < 0)
                            continue;
                        const double* Za = Z + (size_t)36 * ia;
                        for (int a = 0; a < 6; ++a) {
                            double s = 0.0;
                            for (int m = 0; m < 6; ++m)
                                s += Za[6 * m + a] * zz[m];
                            b[6 * fa + a] -= s;
                        }
                        for (int fb = 0; fb <= fa; ++fb) {
                            const int ib = row[fb];
                            if (ib < 0)
                                continue;
                            const double* Zb = Z + (size_t)36 * ib;
                            double* Sblk = S + (size_t)(6 * fa) * nf6 + 6 * fb;
                            for (int a = 0; a < 6; ++a)
                                for (int q = 0; q < 6; ++q) {
                                    double s = 0.0;
                                    for (int m = 0; m < 6; ++m)
                                        s += Za[6 * m + a] * Zb[6 * m + q];
                                    Sblk[(size_t)a * nf6 + q] -= s;
                                }
                        }
                    }
                }
            }
        } else {
            int* fcnt = (int*)calloc((size_t)n_f + 1, sizeof(int));
            for (int i = 0; i < n_obs; ++i)
                fcnt[(elim_tags ? p->obs_cam[i] : p->obs_tag[i]) + 1]++;
            for (int f = 0; f < n_f; ++f)
                fcnt[f + 1] += fcnt[f];
            int* flst = (int*)malloc((size_t)(n_obs > 0 ? n_obs : 1) * sizeof(int));
            int* fpos = (int*)malloc((size_t)(n_f > 0 ? n_f : 1) * sizeof(int));
            memcpy(fpos, fcnt, (size_t)n_f * sizeof(int));
            for (int i = 0; i < n_obs; ++i)
                flst[fpos[elim_tags ? p->obs_cam[i] : p->obs_tag[i]]++] = i;
    #pragma omp parallel for schedule(dynamic, 1)
            for (int fa = 0; fa < n_f; ++fa) {
                for (int ka = fcnt[fa]; ka < fcnt[fa + 1]; ++ka) {
                    const int ia = flst[ka];
                    const int e = elim_tags ? p->obs_tag[ia] : p->obs_cam[ia];
                    const double* zz = ze + 6 * e;
                    const double* Za = Z + (size_t)36 * ia;
                    for (int a = 0; a < 6; ++a) {
                        double s = 0.0;
                        for (int m = 0; m < 6; ++m)
                            s += Za[6 * m + a] * zz[m];
                        b[6 * fa + a] -= s;
                    }
                    for (int kb = cnt[e]; kb < cnt[e + 1]; ++kb) {
                        const int ib = lst[kb];
                        const int fb = elim_tags ? p->obs_cam[ib] : p->obs_tag[ib];
                        if (fb > fa)
                            continue;
                        const double* Zb = Z + (size_t)36 * ib;
                        double* Sblk = S + (size_t)(6 * fa) * nf6 + 6 * fb;
                        for (int a = 0; a < 6; ++a)
                            for (int q = 0; q < 6; ++q) {
                                double s = 0.0;
                                for (int m = 0; m < 6; ++m)
                                    s += Za[6 * m + a] * Zb[6 * m + q];
                                Sblk[(size_t)a * nf6 + q] -= s;
                            }
                    }
                }
            }
            free(fcnt);
            free(flst);
            free(fpos);
        }
        free(obs_of);
        const double t_dbg1 = omp_get_wtime();
        fail = chol_lower(S, nf6, nf6);
        if (getenv("VO_DEBUG_TIMES"))
            fprintf(stderr, "[oracle] Schur product %.1f ms, dense Cholesky %.1f ms\n", 1e3 * (t_dbg1 - t_dbg0),
                    1e3 * (omp_get_wtime() - t_dbg1));
    }
    if (!fail) {
        chol_solve(S, nf6, nf6, b);
        memcpy(y + off_f, b, (size_t)nf6 * sizeof(double));
        /* back-substitution y_e = L_e^{-T} (z_e - sum_f Z_ef y_f) */
#pragma omp parallel for schedule(dynamic, 8)
        for (int e = 0; e < n_e; ++e) {
            double v[6];
            memcpy(v, ze + 6 * e, sizeof(v));
            for (int k = cnt[e]; k < cnt[e + 1]; ++k) {
                const int i = lst[k];
                const int f = elim_tags ? p->obs_cam[i] : p->obs_tag[i];
                const double* Zi = Z + (size_t)36 * i;
                for (int a = 0; a < 6; ++a)
                    for (int q = 0; q < 6; ++q)
                        v[a] -= Zi[6 * a + q] * b[6 * f + q];
            }
            bwd6(Me + 36 * e, v);
            memcpy(y + off_e + 6 * e, v, sizeof(v));
        }
    }
    free(cnt);
    free(lst);
    free(pos);
    free(Me);
    free(ze);
    free(Z);
    free(S);
    free(b);
    return fail;
}

static double active_norm(const work* w, const double* cam_qt, const double* tag_qt)
{
    double s = 0.0;
    for (int c = 0; c < w->n_c; ++c)
        if (w->active[c])
            for (int k = 0; k < 7; ++k)
                s += cam_qt[7 * c + k] * cam_qt[7 * c + k];
    for (int t = 0; t < w->n_t; ++t)
        if (w->active[w->n_c + t])
            for (int k = 0; k < 7; ++k)
                s += tag_qt[7 * t + k] * tag_qt[7 * t + k];
    return sqrt(s);
}

/* |Plus(x, -g) - x| in max- and 2-norm (TrustRegionMinimizer::EvaluateGradientAndJacobian). */
static void gradient_norms(const work* w, const double* cam_qt, const double* tag_qt, double* gmax,
                           double* gnorm)
{
    double mx = 0.0, s2 = 0.0;
    for (int k = 0; k < w->n_pose; ++k) {
        if (!w->active[k])
            continue;
        const double* x = (k < w->n_c) ? cam_qt + 7 * k : tag_qt + 7 * (k - w->n_c);
        double ng[6], xp[7];
        for (int a = 0; a < 6; ++a)
            ng[a] = -w->g[6 * k + a];
        if (k < w->n_c)
            vo_pose_plus(x, ng, xp);
        else
            landmark_plus(w->landmark_points, x, ng, xp);
        for (int a = 0; a < 7; ++a) {
            const double d = fabs(x[a] - xp[a]);
            if (d > mx)
                mx = d;
            s2 += d * d;
        }
    }
    *gmax = mx;
    *gnorm = sqrt(s2);
}

static void push_iter(vo_summary* s, const vo_iteration* it)
{
    if (s->trace && s->iterations < s->trace_capacity)
        s->trace[s->iterations] = *it;
    s->iterations++;
}

int vo_solve(vo_problem* p, const vo_options* o, vo_summary* s)
{
    const double t_begin = now_s();
    const int n_c = p->n_cams, n_t = p->n_tags, n_obs = p->n_obs;
    work w;
    memset(&w, 0, sizeof(w));
    w.n_c = n_c;
    w.n_t = n_t;
    w.n_obs = n_obs;
    w.n_pose = n_c + n_t;
    w.n_tan = 6 * w.n_pose;
    w.landmark_points = p->landmark_points;
#ifdef _OPENMP
    if (o->num_threads > 0)
        omp_set_num_threads(o->num_threads);
#endif
    vo_iteration* user_trace = s->trace;
    const int user_cap = s->trace_capacity;
    memset(s, 0, sizeof(*s));
    s->trace = user_trace;
    s->trace_capacity = user_cap;

    w.J = (double*)malloc((size_t)96 * (n_obs > 0 ? n_obs : 1) * sizeof(double));
    w.r = (double*)malloc((size_t)8 * (n_obs > 0 ? n_obs : 1) * sizeof(double));
    w.g = (double*)calloc((size_t)w.n_tan + 1, sizeof(double));
    w.scale = (double*)malloc(((size_t)w.n_tan + 1) * sizeof(double));
    w.diag = (double*)calloc((size_t)w.n_tan + 1, sizeof(double));
    w.step = (double*)calloc((size_t)w.n_tan + 1, sizeof(double));
    w.delta = (double*)calloc((size_t)w.n_tan + 1, sizeof(double));
    w.active = (int*)calloc((size_t)w.n_pose + 1, sizeof(int));
    double* D2 = (double*)calloc((size_t)w.n_tan + 1, sizeof(double));
    double* cand_c = (double*)malloc((size_t)7 * (n_c > 0 ? n_c : 1) * sizeof(double));
    double* cand_t = (double*)malloc((size_t)7 * (n_t > 0 ? n_t : 1) * sizeof(double));
    for (int k = 0; k < w.n_tan; ++k)
        w.scale[k] = 1.0;
    build_obs_lists(p, &w);

    /* Reduced program: constant blocks (origin tag, TagReconstructor.cpp:669-673) and blocks without
     * residuals are removed; cameras without reconstructed tags are never added (:689-690). */
    for (int i = 0; i < n_obs; ++i) {
        w.active[p->obs_cam[i]] = 1;
        w.active[n_c + p->obs_tag[i]] = 1;
    }
    if (p->fixed_tag >= 0 && p->fixed_tag < n_t)
        w.active[n_c + p->fixed_tag] = 0;
    if (p->landmark_points && p->fixed_tag2 >= 0 && p->fixed_tag2 < n_t)
        w.active[n_c + p->fixed_tag2] = 0;

    int solver = o->linear_solver;
    if (solver == VO_SOLVER_SCHUR_AUTO)
        solver = (n_c >= n_t) ? VO_SOLVER_SCHUR_ELIM_CAMS : VO_SOLVER_SCHUR_ELIM_TAGS;

    double* x_c = p->cam_qt;
    double* x_t = p->tag_qt;
    double x_cost = 0.0, t0;
    int term = VO_NO_CONVERGENCE;

    /* ---- iteration zero ---- */
    vo_iteration it;
    memset(&it, 0, sizeof(it));
    t0 = now_s();
    int bad = eval_full(p, o, &w, x_c, x_t, &x_cost);
    s->num_jacobian_evals++;
    s->time_eval_s += now_s() - t0;
    s->initial_cost = x_cost;
    if (bad) {
        s->termination_type = VO_FAILURE;
        s->final_cost = x_cost;
        goto done;
    }
    if (o->jacobi_scaling) {
        col_sq_norms(p, &w, w.scale);
        for (int k = 0; k < w.n_tan; ++k)
            w.scale[k] = 1.0 / (1.0 + sqrt(w.scale[k]));
        scale_columns(p, &w);
    }
    gradient_norms(&w, x_c, x_t, &it.gradient_max_norm, &(double){ 0 });
    it.iteration = 0;
    it.step_is_valid = 1;
    it.step_is_successful = 1;
    it.cost = x_cost;
    double radius = o->initial_trust_region_radius;
    double decrease_factor = 2.0;
    int reuse_diagonal = 0;
    int num_invalid = 0;
    double x_norm = active_norm(&w, x_c, x_t);
    double last_gmax = it.gradient_max_norm;

    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (it.step_is_successful)
            s->num_successful_steps++;
        else
            s->num_unsuccessful_steps++;
        it.trust_region_radius = radius;
        push_iter(s, &it);
        if (it.iteration >= o->max_num_iterations) {
            term = VO_NO_CONVERGENCE;
            break;
        }
        if (it.step_is_successful && it.gradient_max_norm <= o->gradient_tolerance) {
            term = VO_CONVERGENCE;
            break;
        }
        if (radius <= o->min_trust_region_radius) {
            term = VO_CONVERGENCE;
            break;
        }
        const int iter_no = it.iteration + 1;
        memset(&it, 0, sizeof(it));
        it.iteration = iter_no;

        /* LevenbergMarquardtStrategy::ComputeStep */
        t0 = now_s();
        if (!reuse_diagonal) {
            col_sq_norms(p, &w, w.diag);
            for (int k = 0; k < w.n_tan; ++k) {
                double d = w.diag[k];
                if (d < o->min_lm_diagonal)
                    d = o->min_lm_diagonal;
                if (d > o->max_lm_diagonal)
                    d = o->max_lm_diagonal;
                w.diag[k] = d;
            }
        }
        for (int k = 0; k < w.n_tan; ++k) {
            const double lm = sqrt(w.diag[k] / radius);
            D2[k] = lm * lm;
        }
        int lin_fail;
        if (solver == VO_SOLVER_DENSE_NORMAL)
            lin_fail = solve_dense_normal(p, &w, D2, w.step);
        else
            lin_fail = solve_schur(p, &w, D2, w.step, solver == VO_SOLVER_SCHUR_ELIM_TAGS);
        if (!lin_fail)
            for (int k = 0; k < w.n_tan; ++k)
                if (!isfinite(w.step[k]))
                    lin_fail = 1;
        reuse_diagonal = 1;
        double model_cost_change = 0.0;
        if (!lin_fail) {
            for (int k = 0; k < w.n_tan; ++k)
                w.step[k] = -w.step[k];
            /* model_cost_change = -(J s)^T (r + J s / 2)   (TrustRegionMinimizer::ComputeTrustRegionStep) */
            double* mpart = (double*)malloc((size_t)(n_obs > 0 ? n_obs : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n_obs; ++i) {
                const int c = p->obs_cam[i], t = p->obs_tag[i];
                const double* Jc = w.J + (size_t)96 * i;
                const double* Jt = Jc + 48;
                const double* r = w.r + (size_t)8 * i;
                const double* sc = w.step + 6 * c;
                const double* st = w.step + 6 * (n_c + t);
                double acc = 0.0;
                for (int row = 0; row < 8; ++row) {
                    double m = 0.0;
                    for (int k = 0; k < 6; ++k)
                        m += Jc[6 * row + k] * sc[k] + Jt[6 * row + k] * st[k];
                    acc -= m * (r[row] + m / 2.0);
                }
                mpart[i] = acc;
            }
            for (int i = 0; i < n_obs; ++i)   /* serial, in observation order */
                model_cost_change += mpart[i];
            free(mpart);
        }
        s->time_linear_s += now_s() - t0;
        it.model_cost_change = model_cost_change;
        it.step_is_valid = (!lin_fail && model_cost_change > 0.0);

        if (!it.step_is_valid) {
            /* HandleInvalidStep */
            if (++num_invalid >= o->max_num_consecutive_invalid_steps) {
                term = VO_FAILURE;
                break;
            }
            radius = radius / decrease_factor; /* StepIsInvalid == StepRejected */
            decrease_factor *= 2.0;
            reuse_diagonal = 1;
            it.cost = x_cost;
            it.gradient_max_norm = last_gmax;
            it.step_is_successful = 0;
            continue;
        }
        num_invalid = 0;
        for (int k = 0; k < w.n_tan; ++k)
            w.delta[k] = w.step[k] * w.scale[k];

        /* ComputeCandidatePointAndEvaluateCost */
        for (int c = 0; c < n_c; ++c)
            vo_pose_plus(x_c + 7 * c, w.delta + 6 * c, cand_c + 7 * c);
        for (int t = 0; t < n_t; ++t)
            landmark_plus(p->landmark_points, x_t + 7 * t, w.delta + 6 * (n_c + t), cand_t + 7 * t);
        t0 = now_s();
        double cand_cost = cost_at(p, o, cand_c, cand_t);
        s->num_cost_evals++;
        s->time_eval_s += now_s() - t0;
        if (!isfinite(cand_cost))
            cand_cost = DBL_MAX;

        /* ParameterToleranceReached */
        double sn = 0.0;
        for (int c = 0; c < n_c; ++c)
            if (w.active[c])
                for (int k = 0; k < 7; ++k) {
                    const double d = x_c[7 * c + k] - cand_c[7 * c + k];
                    sn += d * d;
                }
        for (int t = 0; t < n_t; ++t)
            if (w.active[n_c + t])
                for (int k = 0; k < 7; ++k) {
                    const double d = x_t[7 * t + k] - cand_t[7 * t + k];
                    sn += d * d;
                }
        it.step_norm = sqrt(sn);
        if (it.step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
            term = VO_CONVERGENCE;
            break;
        }
        /* FunctionToleranceReached */
        it.cost_change = x_cost - cand_cost;
        if (fabs(it.cost_change) <= o->function_tolerance * x_cost) {
            term = VO_CONVERGENCE;
            break;
        }
        /* IsStepSuccessful (monotonic: step quality == cost_change / model_cost_change) */
        it.relative_decrease = (cand_cost >= DBL_MAX) ? -DBL_MAX : it.cost_change / model_cost_change;
        if (it.relative_decrease > o->min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(x_c, cand_c, (size_t)7 * n_c * sizeof(double));
            memcpy(x_t, cand_t, (size_t)7 * n_t * sizeof(double));
            x_norm = active_norm(&w, x_c, x_t);
            t0 = now_s();
            bad = eval_full(p, o, &w, x_c, x_t, &x_cost);
            s->num_jacobian_evals++;
            s->time_eval_s += now_s() - t0;
            if (bad) {
                term = VO_FAILURE;
                break;
            }
            if (o->jacobi_scaling)
                scale_columns(p, &w);
            gradient_norms(&w, x_c, x_t, &it.gradient_max_norm, &(double){ 0 });
            last_gmax = it.gradient_max_norm;
            it.step_is_successful = 1;
            it.cost = x_cost;
            /* LevenbergMarquardtStrategy::StepAccepted */
            {
                const double q = 2.0 * it.relative_decrease - 1.0;
                double den = 1.0 - q * q * q;
                if (den < 1.0 / 3.0)
                    den = 1.0 / 3.0;
                radius = radius / den;
                if (radius > o->max_trust_region_radius)
                    radius = o->max_trust_region_radius;
                decrease_factor = 2.0;
                reuse_diagonal = 0;
            }
        } else {
            /* HandleUnsuccessfulStep / StepRejected */
            it.step_is_successful = 0;
            it.cost = cand_cost;
            it.gradient_max_norm = last_gmax;
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = 1;
        }
    }
    s->termination_type = term;
    s->final_cost = x_cost;

done:
    free(w.J);
    free(w.r);
    free(w.g);
    free(w.scale);
    free(w.diag);
    free(w.step);
    free(w.delta);
    free(w.active);
    free(w.off);
    free(w.lst);
    free(D2);
    free(cand_c);
    free(cand_t);
    s->time_total_s = now_s() - t_begin;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Reprojection statistics (TagReconstructor.cpp:340-455)
 * ---------------------------------------------------------------------------------------------- */

/* Eigen::Quaterniond::toRotationMatrix (no normalisation), as used at TagReconstructor.cpp:356 and
 * TagReconstructor.h:37. */
static void eigen_quat_to_R(const double q[4], double R[9])
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz);
    R[1] = txy - twz;
    R[2] = txz + twy;
    R[3] = txy + twz;
    R[4] = 1.0 - (txx + tzz);
    R[5] = tyz - twx;
    R[6] = txz - twy;
    R[7] = tyz + twx;
    R[8] = 1.0 - (txx + tyy);
}

void vo_reprojection_stats(const vo_problem* p, double* per_cam_mean, double* per_tag_mean,
                           double* avg, double* per_corner)
{
    const int n_c = p->n_cams, n_t = p->n_tags;
    double* sc = (double*)calloc((size_t)n_c + 1, sizeof(double));
    double* st = (double*)calloc((size_t)n_t + 1, sizeof(double));
    int* nc = (int*)calloc((size_t)n_c + 1, sizeof(int));
    int* nt = (int*)calloc((size_t)n_t + 1, sizeof(int));
    for (int i = 0; i < p->n_obs; ++i) {
        const int c = p->obs_cam[i], t = p->obs_tag[i];
        const double* cq = p->cam_qt + 7 * c;
        const double* tq = p->tag_qt + 7 * t;
        double Rc[9], Rt[9];
        eigen_quat_to_R(cq, Rc);
        eigen_quat_to_R(tq, Rt);
        double sum = 0.0;
        for (int k = 0; k < 4; ++k) {
            double cl[3], pw[3], pc[3], uv[2];
            local_corner(p->tag_wh + 2 * t, k, cl);
            for (int a = 0; a < 3; ++a)  /* TagReconstructor.h:40 */
                pw[a] = (Rt[3 * a] * cl[0] + Rt[3 * a + 1] * cl[1] + Rt[3 * a + 2] * cl[2]) + tq[4 + a];
            for (int a = 0; a < 3; ++a)  /* TagReconstructor.cpp:362 */
                pc[a] = (Rc[3 * a] * pw[0] + Rc[3 * a + 1] * pw[1] + Rc[3 * a + 2] * pw[2]) + cq[4 + a];
            project_camera_model(p->intr, p->dist, pc, uv);   /* :362: camModel.projectPoint */
            const double du = uv[0] - p->obs_px[8 * i + 2 * k];
            const double dv = uv[1] - p->obs_px[8 * i + 2 * k + 1];
            if (per_corner) {
                per_corner[8 * i + 2 * k] = du;
                per_corner[8 * i + 2 * k + 1] = dv;
            }
            sum += sqrt(du * du + dv * dv);
        }
        sc[c] += sum;
        st[t] += sum;
        nc[c] += 4;
        nt[t] += 4;
    }
    if (per_cam_mean)
        for (int c = 0; c < n_c; ++c)
            per_cam_mean[c] = nc[c] ? sc[c] / nc[c] : -1.0;
    double a = 0.0;
    long tot = 0;
    for (int t = 0; t < n_t; ++t) {
        if (nt[t]) {
            a += st[t];
            tot += nt[t];
        }
        if (per_tag_mean)
            per_tag_mean[t] = nt[t] ? st[t] / nt[t] : NAN;
    }
    if (avg)
        *avg = tot ? a / (double)tot : 0.0;
    free(sc);
    free(st);
    free(nc);
    free(nt);
}

/* ------------------------------------------------------------------------------------------------
 * Covariance of the tag translations (TagReconstructor.cpp:744-783: ceres::Covariance on the (t, t)
 * blocks of every reconstructed tag): the 3x3 diagonal blocks of (J^T J)^-1 in tangent coordinates,
 * J corrected by the loss when robustify is set (Covariance::Options::apply_loss_function = true).
 * Constant / residual-free blocks get zeros.  Dense Cholesky of the full normal matrix -- small cases.
 * Returns nonzero when J^T J is not positive definite.
 * ---------------------------------------------------------------------------------------------- */
int vo_tag_translation_covariance(const vo_problem* p, const vo_options* o, double* cov)
{
    const int n_c = p->n_cams, n_t = p->n_tags, n_obs = p->n_obs;
    work w;
    memset(&w, 0, sizeof(w));
    w.n_c = n_c;
    w.n_t = n_t;
    w.n_obs = n_obs;
    w.n_pose = n_c + n_t;
    w.n_tan = 6 * w.n_pose;
    const int n = w.n_tan;
    w.J = (double*)malloc((size_t)96 * (n_obs > 0 ? n_obs : 1) * sizeof(double));
    w.r = (double*)malloc((size_t)8 * (n_obs > 0 ? n_obs : 1) * sizeof(double));
    w.g = (double*)calloc((size_t)n + 1, sizeof(double));
    w.active = (int*)calloc((size_t)w.n_pose + 1, sizeof(int));
    build_obs_lists(p, &w);
    for (int i = 0; i < n_obs; ++i) {
        w.active[p->obs_cam[i]] = 1;
        w.active[n_c + p->obs_tag[i]] = 1;
    }
    if (p->fixed_tag >= 0 && p->fixed_tag < n_t)
        w.active[n_c + p->fixed_tag] = 0;
    if (p->landmark_points && p->fixed_tag2 >= 0 && p->fixed_tag2 < n_t)
        w.active[n_c + p->fixed_tag2] = 0;
    double cost = 0.0;
    eval_full(p, o, &w, p->cam_qt, p->tag_qt, &cost);
    double* D2 = (double*)calloc((size_t)n + 1, sizeof(double));
    for (int q = 0; q < w.n_pose; ++q)
        for (int k = 0; k < 6; ++k)
            D2[6 * q + k] = w.active[q] ? 0.0 : 1.0;
    /* H = J^T J (+ unit diagonal on inactive blocks), factored once */
    double* H = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n_obs; ++i) {
        const int c = p->obs_cam[i], t = p->obs_tag[i];
        const double* Jc = w.J + (size_t)96 * i;
        const double* Jt = Jc + 48;
        const int ic = 6 * c, it = 6 * (n_c + t);
        for (int row = 0; row < 8; ++row) {
            const double* jc = Jc + 6 * row;
            const double* jt = Jt + 6 * row;
            for (int a = 0; a < 6; ++a) {
                for (int q = 0; q <= a; ++q) {
                    H[(size_t)(ic + a) * n + ic + q] += jc[a] * jc[q];
                    H[(size_t)(it + a) * n + it + q] += jt[a] * jt[q];
                }
                for (int q = 0; q < 6; ++q)
                    H[(size_t)(it + a) * n + ic + q] += jt[a] * jc[q];
            }
        }
    }
    for (int i = 0; i < n; ++i)
        H[(size_t)i * n + i] += D2[i];
    int fail = chol_lower(H, n, n);
    double* x = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    for (int t = 0; t < n_t; ++t) {
        double* out = cov + 9 * t;
        if (fail || !w.active[n_c + t]) {
            for (int k = 0; k < 9; ++k)
                out[k] = 0.0;
            continue;
        }
        const int base = 6 * (n_c + t);
        for (int a = 0; a < 3; ++a) {
            memset(x, 0, (size_t)n * sizeof(double));
            x[base + a] = 1.0;
            chol_solve(H, n, n, x);
            for (int b = 0; b < 3; ++b)
                out[3 * b + a] = x[base + b];
        }
    }
    free(x);
    free(H);
    free(D2);
    free(w.J);
    free(w.r);
    free(w.g);
    free(w.active);
    free(w.off);
    free(w.lst);
    return fail;
}
