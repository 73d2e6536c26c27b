// Per-corner reprojection arithmetic shared by the evaluation kernels (gfx950 device code).
//
// Reference being replaced (paths relative to /root/reference):
//   residual   include/visual_marker_mapping/TagReconstructionCostFunction.h:101-159
//   projection src/CameraModel.cpp:6-26
//   quad       include/visual_marker_mapping/TagReconstructor.h:44-52
// The tangent Jacobians are what AutoDiffCostFunction<...,2,3,4,3,4> (CostFunction.h:167) followed by
// ceres::QuaternionParameterization (src/TagReconstructor.cpp:661) produce; derivation in DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace vmm {

struct Intrinsics {
    double fx, fy, cx, cy;
    double k1, k2, p1, p2, k3;
};

// Rotation + translation of one pose, expanded once per thread.
struct Rigid {
    double R[9];
    double t[3];
};

// NORMALIZE=true: ceres::QuaternionRotatePoint semantics (q / |q| first, CostFunction.h:109,118).
// NORMALIZE=false: Eigen::Quaterniond::toRotationMatrix as used by the statistics path
// (src/TagReconstructor.cpp:356, TagReconstructor.h:37).
template <bool NORMALIZE>
__device__ __forceinline__ void load_rigid(const double* __restrict__ qt, Rigid& o)
{
    double w = qt[0], x = qt[1], y = qt[2], z = qt[3];
    if (NORMALIZE) {
        const double n = 1.0 / sqrt(w * w + x * x + y * y + z * z);
        w *= n;
        x *= n;
        y *= n;
        z *= n;
    }
    o.R[0] = 1.0 - 2.0 * (y * y + z * z);
    o.R[1] = 2.0 * (x * y - w * z);
    o.R[2] = 2.0 * (x * z + w * y);
    o.R[3] = 2.0 * (x * y + w * z);
    o.R[4] = 1.0 - 2.0 * (x * x + z * z);
    o.R[5] = 2.0 * (y * z - w * x);
    o.R[6] = 2.0 * (x * z - w * y);
    o.R[7] = 2.0 * (y * z + w * x);
    o.R[8] = 1.0 - 2.0 * (x * x + y * y);
    o.t[0] = qt[4];
    o.t[1] = qt[5];
    o.t[2] = qt[6];
}

// Everything one corner contributes: residual (2) and, when requested, the 2x6 Jacobian rows with
// respect to the camera tangent (t, delta) and the tag tangent (t, delta).
struct CornerEval {
    double ru, rv;
    double jc[2][6];
    double jt[2][6];
};

// The tangential y term of the two projections the reference holds:
//   functor      (CostFunction.h:143-144)  yd = y rad + 2 p2 x  y + p1 (r2 + 2 y^2)
//   CameraModel  (src/CameraModel.cpp:20-23) pt.x() has already been overwritten with the distorted xd when
//                pt.y() is computed:        yd = y rad + 2 p2 xd y + p1 (r2 + 2 y^2)
// The cost/Jacobian path is the functor; the statistics and projectPoint are CameraModel (they differ by up to
// 0.02 px with the README distortion).
__device__ __forceinline__ void distort(const Intrinsics& K, bool camera_model, double x, double y, double r2,
                                        double rad, double& xd, double& yd)
{
    xd = x * rad + 2.0 * K.p1 * x * y + K.p2 * (r2 + 2.0 * x * x);
    yd = y * rad + 2.0 * K.p2 * (camera_model ? xd : x) * y + K.p1 * (r2 + 2.0 * y * y);
}

// sx, sy in {-1,+1}: which corner of the quad; hw, hh: half width / half height of the tag.
// CAMERA_MODEL: project like CameraModel::projectPoint (statistics path; no Jacobians exist for it).
template <bool NEED_JC, bool NEED_JT, bool CAMERA_MODEL = false>
__device__ __forceinline__ void eval_corner(const Intrinsics& K, const Rigid& cam, const Rigid& tag,
                                            double sxhw, double syhh, double u_obs, double v_obs,
                                            CornerEval& e, const bool on = true)
{
    // a = R_t p_l (p_l.z == 0), P_w = a + t_t                        CostFunction.h:107-114
    const double a0 = tag.R[0] * sxhw + tag.R[1] * syhh;
    const double a1 = tag.R[3] * sxhw + tag.R[4] * syhh;
    const double a2 = tag.R[6] * sxhw + tag.R[7] * syhh;
    const double w0 = a0 + tag.t[0], w1 = a1 + tag.t[1], w2 = a2 + tag.t[2];
    // b = R_c P_w, P_c = b + t_c                                      CostFunction.h:117-122
    const double b0 = cam.R[0] * w0 + cam.R[1] * w1 + cam.R[2] * w2;
    const double b1 = cam.R[3] * w0 + cam.R[4] * w1 + cam.R[5] * w2;
    const double b2 = cam.R[6] * w0 + cam.R[7] * w1 + cam.R[8] * w2;
    const double X = b0 + cam.t[0], Y = b1 + cam.t[1], Z = b2 + cam.t[2];
    // a switched-off observation (on == false; weighted zero by the caller) is evaluated at the principal
    // point instead: its poses are parked defaults and may put the corner on the camera plane (Z == 0), and
    // 0 * inf would poison the sums
    const double iz = on ? 1.0 / Z : 0.0;
    const double x = X * iz, y = Y * iz;             // :125-126
    const double r2 = x * x + y * y;                 // :129
    const double rad = 1.0 + r2 * (K.k1 + r2 * (K.k2 + r2 * K.k3));
    static_assert(!(CAMERA_MODEL && (NEED_JC || NEED_JT)), "the CameraModel projection is residual-only");
    double xd, yd;
    distort(K, CAMERA_MODEL, x, y, r2, rad, xd, yd);   // :141-144 / CameraModel.cpp:20-23
    e.ru = K.fx * xd + K.cx - u_obs;                 // :151-156
    e.rv = K.fy * yd + K.cy - v_obs;
    if (!NEED_JC && !NEED_JT)
        return;
    const double dr = K.k1 + r2 * (2.0 * K.k2 + 3.0 * K.k3 * r2);
    const double D00 = rad + 2.0 * x * x * dr + 2.0 * K.p1 * y + 6.0 * K.p2 * x;
    const double D01 = 2.0 * x * y * dr + 2.0 * K.p1 * x + 2.0 * K.p2 * y;
    const double D11 = rad + 2.0 * y * y * dr + 2.0 * K.p2 * x + 6.0 * K.p1 * y;
    // G = diag(fx,fy) D [iz 0 -x iz; 0 iz -y iz]   (d residual / d P_c)
    double g[2][3];
    g[0][0] = K.fx * D00 * iz;
    g[0][1] = K.fx * D01 * iz;
    g[0][2] = -K.fx * (D00 * x + D01 * y) * iz;
    g[1][0] = K.fy * D01 * iz;
    g[1][1] = K.fy * D11 * iz;
    g[1][2] = -K.fy * (D01 * x + D11 * y) * iz;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double g0 = g[r][0], g1 = g[r][1], g2 = g[r][2];
        if (NEED_JC) {
            e.jc[r][0] = g0;
            e.jc[r][1] = g1;
            e.jc[r][2] = g2;
            // -2 g^T [b]x = 2 (b x g): rotation about the camera-frame point, half-angle tangent
            e.jc[r][3] = 2.0 * (b1 * g2 - b2 * g1);
            e.jc[r][4] = 2.0 * (b2 * g0 - b0 * g2);
            e.jc[r][5] = 2.0 * (b0 * g1 - b1 * g0);
        }
        if (NEED_JT) {
            const double h0 = g0 * cam.R[0] + g1 * cam.R[3] + g2 * cam.R[6];
            const double h1 = g0 * cam.R[1] + g1 * cam.R[4] + g2 * cam.R[7];
            const double h2 = g0 * cam.R[2] + g1 * cam.R[5] + g2 * cam.R[8];
            e.jt[r][0] = h0;
            e.jt[r][1] = h1;
            e.jt[r][2] = h2;
            e.jt[r][3] = 2.0 * (a1 * h2 - a2 * h1);
            e.jt[r][4] = 2.0 * (a2 * h0 - a0 * h2);
            e.jt[r][5] = 2.0 * (a0 * h1 - a1 * h0);
        }
    }
}

// Point-landmark functor: OpenCVReprojectionError::operator() (CostFunction.h:21-68), the residual of the dead
// doBundleAdjustment_points (src/TagReconstructor.cpp:457-644).  The camera rotation is
// ceres::UnitQuaternionRotatePoint (:27): the polynomial R_u(q) = I + 2 M(q) of the quaternion AS IT IS, no
// normalisation (cam.R comes from load_rigid<false>).  Jacobians as AutoDiffCostFunction<..., 2, 3, 4, 3> (:74-75)
// + QuaternionParameterization (src/TagReconstructor.cpp:529) give them: d/d(point) = A R_u(q) and
// d/d(delta) = A [d R_u(q) p / dq] G(q), G = the 4x3 Jacobian of Plus at delta = 0.
// e.jt[r][0..2] receives the point Jacobian (3 columns); e.jt[r][3..5] is not written.
template <bool NEED_JC, bool NEED_JP>
__device__ __forceinline__ void eval_point(const Intrinsics& K, const Rigid& cam, const double* __restrict__ camq,
                                           const double p0, const double p1, const double p2, double u_obs,
                                           double v_obs, CornerEval& e, const bool on = true)
{
    const double b0 = cam.R[0] * p0 + cam.R[1] * p1 + cam.R[2] * p2;
    const double b1 = cam.R[3] * p0 + cam.R[4] * p1 + cam.R[5] * p2;
    const double b2 = cam.R[6] * p0 + cam.R[7] * p1 + cam.R[8] * p2;
    const double X = b0 + cam.t[0], Y = b1 + cam.t[1], Z = b2 + cam.t[2];   // :27-31
    const double iz = on ? 1.0 / Z : 0.0;
    const double x = X * iz, y = Y * iz;             // :34-35
    const double r2 = x * x + y * y;                 // :38
    const double rad = 1.0 + r2 * (K.k1 + r2 * (K.k2 + r2 * K.k3));
    double xd, yd;
    distort(K, false, x, y, r2, rad, xd, yd);        // :49-54
    e.ru = K.fx * xd + K.cx - u_obs;                 // :61-67
    e.rv = K.fy * yd + K.cy - v_obs;
    if (!NEED_JC && !NEED_JP)
        return;
    const double dr = K.k1 + r2 * (2.0 * K.k2 + 3.0 * K.k3 * r2);
    const double D00 = rad + 2.0 * x * x * dr + 2.0 * K.p1 * y + 6.0 * K.p2 * x;
    const double D01 = 2.0 * x * y * dr + 2.0 * K.p1 * x + 2.0 * K.p2 * y;
    const double D11 = rad + 2.0 * y * y * dr + 2.0 * K.p2 * x + 6.0 * K.p1 * y;
    double g[2][3];
    g[0][0] = K.fx * D00 * iz;
    g[0][1] = K.fx * D01 * iz;
    g[0][2] = -K.fx * (D00 * x + D01 * y) * iz;
    g[1][0] = K.fy * D01 * iz;
    g[1][1] = K.fy * D11 * iz;
    g[1][2] = -K.fy * (D01 * x + D11 * y) * iz;
    double dk[3][3];   // d(R_u p)/d(delta): [component][tangent direction]
    if (NEED_JC) {
        const double qw = camq[0], qx = camq[1], qy = camq[2], qz = camq[3];
        // columns of d(R_u p)/d(w, x, y, z)
        const double dq[3][4] = {
            { 2.0 * (qy * p2 - qz * p1), 2.0 * (qy * p1 + qz * p2), 2.0 * (-2.0 * qy * p0 + qx * p1 + qw * p2),
              2.0 * (-2.0 * qz * p0 - qw * p1 + qx * p2) },
            { 2.0 * (qz * p0 - qx * p2), 2.0 * (qy * p0 - 2.0 * qx * p1 - qw * p2), 2.0 * (qx * p0 + qz * p2),
              2.0 * (qw * p0 - 2.0 * qz * p1 + qy * p2) },
            { 2.0 * (qx * p1 - qy * p0), 2.0 * (qz * p0 + qw * p1 - 2.0 * qx * p2), 2.0 * (-qw * p0 + qz * p1 - 2.0 * qy * p2),
              2.0 * (qx * p0 + qy * p1) } };
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dk[a][0] = -dq[a][0] * qx + dq[a][1] * qw - dq[a][2] * qz + dq[a][3] * qy;
            dk[a][1] = -dq[a][0] * qy + dq[a][1] * qz + dq[a][2] * qw - dq[a][3] * qx;
            dk[a][2] = -dq[a][0] * qz - dq[a][1] * qy + dq[a][2] * qx + dq[a][3] * qw;
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double g0 = g[r][0], g1 = g[r][1], g2 = g[r][2];
        if (NEED_JC) {
            e.jc[r][0] = g0;
            e.jc[r][1] = g1;
            e.jc[r][2] = g2;
            e.jc[r][3] = g0 * dk[0][0] + g1 * dk[1][0] + g2 * dk[2][0];
            e.jc[r][4] = g0 * dk[0][1] + g1 * dk[1][1] + g2 * dk[2][1];
            e.jc[r][5] = g0 * dk[0][2] + g1 * dk[1][2] + g2 * dk[2][2];
        }
        if (NEED_JP) {
            e.jt[r][0] = g0 * cam.R[0] + g1 * cam.R[3] + g2 * cam.R[6];
            e.jt[r][1] = g0 * cam.R[1] + g1 * cam.R[4] + g2 * cam.R[7];
            e.jt[r][2] = g0 * cam.R[2] + g1 * cam.R[5] + g2 * cam.R[8];
        }
    }
}

// ceres::HuberLoss(a)::Evaluate on s = |r|^2 (src/TagReconstructor.cpp:721): returns rho(s) and the
// row weight sqrt(rho'(s)) the Ceres corrector applies when rho'' <= 0.
__device__ __forceinline__ void huber(bool robust, double a, double s, double& rho0, double& wgt)
{
    rho0 = s;
    wgt = 1.0;
    if (robust && s > a * a) {
        const double r = sqrt(s);
        rho0 = 2.0 * a * r - a * a;
        double rho1 = a / r;
        rho1 = rho1 < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : rho1;
        wgt = sqrt(rho1);
    }
}

// ceres::QuaternionParameterization::Plus on the rotation part, plain addition on the translation.
__device__ __forceinline__ void pose_plus(const double* __restrict__ qt, const double d[6], double out[7])
{
    out[4] = qt[4] + d[0];
    out[5] = qt[5] + d[1];
    out[6] = qt[6] + d[2];
    const double nd = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (nd > 0.0) {
        const double s = sin(nd) / nd;
        const double z0 = cos(nd), z1 = s * d[3], z2 = s * d[4], z3 = s * d[5];
        const double w0 = qt[0], w1 = qt[1], w2 = qt[2], w3 = qt[3];
        out[0] = z0 * w0 - z1 * w1 - z2 * w2 - z3 * w3;
        out[1] = z0 * w1 + z1 * w0 + z2 * w3 - z3 * w2;
        out[2] = z0 * w2 - z1 * w3 + z2 * w0 + z3 * w1;
        out[3] = z0 * w3 + z1 * w2 - z2 * w1 + z3 * w0;
    } else {
        out[0] = qt[0];
        out[1] = qt[1];
        out[2] = qt[2];
        out[3] = qt[3];
    }
}

// Butterfly sum over the 64 lanes of a wave; every lane ends with the total.  Fixed tree, so the
// result does not depend on scheduling.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// Sums 32 per-lane values over the wave with ONE butterfly for all of them: at every level a lane keeps
// half of its values and sends the other half to its partner (lane ^ offset), so the number of live
// values halves while the span doubles -- 31 exchanges instead of 32 x 6.  Returns the total of value
// wave_sum32_index(lane); lanes 2m and 2m+1 hold the same value.  Fixed tree: deterministic.
__device__ __forceinline__ double wave_sum32(double (&v)[32], const int lane)
{
#pragma unroll
    for (int level = 0; level < 5; ++level) {
        const int off = 32 >> level;
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < (16 >> level); ++j) {
            const double keep = hi ? v[2 * j + 1] : v[2 * j];
            const double send = hi ? v[2 * j] : v[2 * j + 1];
            v[j] = keep + __shfl_xor(send, off, 64);
        }
    }
    return v[0] + __shfl_xor(v[0], 1, 64);
}

__device__ __forceinline__ int wave_sum32_index(const int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3)
        | (((lane >> 1) & 1) << 4);
}

} // namespace vmm
