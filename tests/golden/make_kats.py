#!/usr/bin/env python3
"""Generates tests/golden/kat_residual.json: 50-digit mpmath known answers for the hot path's
per-corner arithmetic.  Written from the reference's formulas, independently of oracle/ and of the
HIP code:

  residual        /root/reference/include/visual_marker_mapping/TagReconstructionCostFunction.h:101-159
  corner quad     /root/reference/include/visual_marker_mapping/TagReconstructor.h:44-52
  Plus            ceres::QuaternionParameterization (call site src/TagReconstructor.cpp:661)
  Huber           ceres::HuberLoss(1.0)            (call site src/TagReconstructor.cpp:721)

Tangent Jacobians are d residual(Plus(x, d)) / d d at d = 0, differentiated numerically by mpmath
at 50 digits (columns: translation(3) then half-angle rotation(3), camera block then tag block).
The reference itself holds no golden vectors, so these are the committed fixtures.

Run:  python tests/golden/make_kats.py   (needs mpmath; deterministic)
"""
import json
import os
import random

from mpmath import mp, mpf, sqrt, sin, cos, diff

mp.dps = 50


def quat_rotate(q, p):
    n = sqrt(sum(c * c for c in q))
    w, x, y, z = [c / n for c in q]
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
    return [sum(R[i][k] * p[k] for k in range(3)) for i in range(3)]


def plus(qt, d):
    q, t = qt[:4], qt[4:]
    t2 = [t[i] + d[i] for i in range(3)]
    dv = d[3:6]
    nd = sqrt(sum(c * c for c in dv))
    if nd == 0:
        return list(q) + t2
    s = sin(nd) / nd
    z = [cos(nd), s * dv[0], s * dv[1], s * dv[2]]
    w = q
    q2 = [z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3],
          z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2],
          z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1],
          z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0]]
    return q2 + t2


def corner_residual(intr, dist, cam, tag, cl, uv):
    pw = quat_rotate(tag[:4], cl)
    pw = [pw[i] + tag[4 + i] for i in range(3)]
    pc = quat_rotate(cam[:4], pw)
    pc = [pc[i] + cam[4 + i] for i in range(3)]
    x, y = pc[0] / pc[2], pc[1] / pc[2]
    r2 = x * x + y * y
    k1, k2, p1, p2, k3 = dist
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    return [intr[0] * xd + intr[2] - uv[0], intr[1] * yd + intr[3] - uv[1]]


def unit_quat_rotate(q, p):
    """ceres::UnitQuaternionRotatePoint: the rotation polynomial of q WITHOUT normalisation
    (TagReconstructionCostFunction.h:27, OpenCVReprojectionError)."""
    w, x, y, z = q
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
    return [sum(R[i][k] * p[k] for k in range(3)) for i in range(3)]


def point_residual(intr, dist, cam, pt, uv):
    """OpenCVReprojectionError::operator(), TagReconstructionCostFunction.h:21-68."""
    pc = unit_quat_rotate(cam[:4], pt)
    pc = [pc[i] + cam[4 + i] for i in range(3)]
    x, y = pc[0] / pc[2], pc[1] / pc[2]
    r2 = x * x + y * y
    k1, k2, p1, p2, k3 = dist
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    return [intr[0] * xd + intr[2] - uv[0], intr[1] * yd + intr[3] - uv[1]]


def project_camera_model(intr, dist, pc):
    """CameraModel::projectPoint, /root/reference/src/CameraModel.cpp:6-26, statement by statement: pt.x() is
    overwritten at :20-21 BEFORE :22-23 evaluates 2*p2*pt.x()*pt.y(), so the y term sees the DISTORTED x
    (r2 and pt.y() are still the undistorted values).  Not the functor's formula (CostFunction.h:141-144)."""
    x, y = pc[0] / pc[2], pc[1] / pc[2]
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    xd = x * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) + 2 * p2 * xd * y + p1 * (r2 + 2 * y * y)
    return [intr[0] * xd + intr[2], intr[1] * yd + intr[3]]


def eigen_rotate(q, p):
    """Eigen::Quaterniond::toRotationMatrix() * p -- no normalisation (TagReconstructor.h:37,
    src/TagReconstructor.cpp:356)."""
    w, x, y, z = q
    R = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
    return [sum(R[i][k] * p[k] for k in range(3)) for i in range(3)]


def corner_reprojection_error(intr, dist, cam, tag, cl, uv):
    """One term of computeReprojectionErrorPer{Img,Tag,Corner} (src/TagReconstructor.cpp:353-363):
    camModel.projectPoint(R_c * (R_t * corner + t_t) + t_c) - observed corner."""
    pw = eigen_rotate(tag[:4], cl)
    pw = [pw[i] + tag[4 + i] for i in range(3)]
    pc = eigen_rotate(cam[:4], pw)
    pc = [pc[i] + cam[4 + i] for i in range(3)]
    uvp = project_camera_model(intr, dist, pc)
    return [uvp[0] - uv[0], uvp[1] - uv[1]]


def local_corners(w, h):
    return [[-w / 2, -h / 2, mpf(0)], [w / 2, -h / 2, mpf(0)], [w / 2, h / 2, mpf(0)],
            [-w / 2, h / 2, mpf(0)]]


def rand_quat(rng, small=False):
    if small:
        v = [1.0] + [rng.gauss(0, 0.08) for _ in range(3)]
    else:
        v = [rng.gauss(0, 1) for _ in range(4)]
    n = sum(c * c for c in v) ** 0.5
    return [c / n for c in v]


def main():
    rng = random.Random(20261004)
    intr = [8.0752937867635346e+03, 8.0831676114192869e+03, 3.0163896805084278e+03,
            1.9962896554785455e+03]
    dist_readme = [-1.8618183262669760e-01, 3.7018092365577054e-01, -2.9390604003594177e-04,
                   4.1533180829908799e-04, 5.7043887874185996e-02]
    cases = []
    for ci in range(12):
        dist = [0.0] * 5 if ci % 2 == 0 else dist_readme
        # tag near the origin plane, camera ~3 m in front looking roughly at it
        tag = rand_quat(rng, small=True) + [rng.uniform(-1, 1), rng.uniform(-0.6, 0.6),
                                            rng.gauss(0, 0.05)]
        # camera: rotate 180 deg about x (looks down -z with y down) times a small rotation
        qs = rand_quat(rng, small=True)
        qx = [0.0, 1.0, 0.0, 0.0]
        cam_q = [qs[0] * qx[0] - qs[1] * qx[1] - qs[2] * qx[2] - qs[3] * qx[3],
                 qs[0] * qx[1] + qs[1] * qx[0] + qs[2] * qx[3] - qs[3] * qx[2],
                 qs[0] * qx[2] - qs[1] * qx[3] + qs[2] * qx[0] + qs[3] * qx[1],
                 qs[0] * qx[3] + qs[1] * qx[2] - qs[2] * qx[1] + qs[3] * qx[0]]
        if ci in (4, 5):  # non-unit quaternions exercise the normalisation inside the functor
            cam_q = [c * 1.7 for c in cam_q]
            tag[:4] = [c * 0.6 for c in tag[:4]]
        cam = cam_q + [rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4), rng.uniform(2.5, 4.0)]
        w, h = (0.1285, 0.1285) if ci % 3 else (0.1165, 0.0923)
        camm, tagm = [mpf(c) for c in cam], [mpf(c) for c in tag]
        intrm, distm = [mpf(c) for c in intr], [mpf(c) for c in dist]
        cls = local_corners(mpf(w), mpf(h))
        # observation = exact projection + a few px offset so residuals are O(1)
        px = []
        for cl in cls:
            r0 = corner_residual(intrm, distm, camm, tagm, cl, [mpf(0), mpf(0)])
            px += [float(r0[0]) + rng.gauss(0, 1.5), float(r0[1]) + rng.gauss(0, 1.5)]
        pxm = [mpf(c) for c in px]
        res, Jc, Jt = [], [], []
        for k, cl in enumerate(cls):
            uv = pxm[2 * k:2 * k + 2]
            r = corner_residual(intrm, distm, camm, tagm, cl, uv)
            res += [r[0], r[1]]
            for comp in range(2):
                rowc, rowt = [], []
                for a in range(6):
                    def fc(e, a=a, comp=comp):
                        d = [mpf(0)] * 6
                        d[a] = e
                        return corner_residual(intrm, distm, plus(camm, d), tagm, cl, uv)[comp]

                    def ft(e, a=a, comp=comp):
                        d = [mpf(0)] * 6
                        d[a] = e
                        return corner_residual(intrm, distm, camm, plus(tagm, d), cl, uv)[comp]
                    rowc.append(diff(fc, mpf(0)))
                    rowt.append(diff(ft, mpf(0)))
                Jc.append(rowc)
                Jt.append(rowt)
        # the same observation through the STATISTICS path (CameraModel::projectPoint + Eigen rotations)
        rep = []
        for k, cl in enumerate(cls):
            rep += corner_reprojection_error(intrm, distm, camm, tagm, cl, pxm[2 * k:2 * k + 2])
        cases.append({
            "intr": intr, "dist": dist, "cam_qt": cam, "tag_qt": tag, "wh": [w, h], "px": px,
            "residual": [float(v) for v in res],
            "reprojection_error_camera_model": [float(v) for v in rep],
            "J_cam": [[float(v) for v in row] for row in Jc],
            "J_tag": [[float(v) for v in row] for row in Jt],
        })
    # Plus known answers
    plus_cases = []
    for _ in range(6):
        qt = rand_quat(rng) + [rng.uniform(-2, 2) for _ in range(3)]
        d = [rng.gauss(0, 0.3) for _ in range(6)]
        out = plus([mpf(c) for c in qt], [mpf(c) for c in d])
        plus_cases.append({"qt": qt, "delta": d, "out": [float(v) for v in out]})
    plus_cases.append({"qt": plus_cases[0]["qt"], "delta": [0.1, -0.2, 0.3, 0.0, 0.0, 0.0],
                       "out": [float(v) for v in plus([mpf(c) for c in plus_cases[0]["qt"]],
                                                      [mpf("0.1"), mpf("-0.2"), mpf("0.3"), mpf(0),
                                                       mpf(0), mpf(0)])]})
    # Huber(a=1) known answers: rho, rho', rho''
    huber = []
    for s in [0.0, 0.25, 1.0, 1.0000001, 2.0, 9.0, 400.0, 1e6]:
        sm = mpf(s)
        if sm > 1:
            r = sqrt(sm)
            rho = [2 * r - 1, 1 / r, -(1 / r) / (2 * sm)]
        else:
            rho = [sm, mpf(1), mpf(0)]
        huber.append({"a": 1.0, "s": s, "rho": [float(v) for v in rho]})
    # CameraModel::projectPoint known answers (aliased y term), README distortion and none, normalised image
    # coordinates out to |x|, |y| ~ 0.4 (the image corners of the README camera are at 0.37 / 0.25)
    rng2 = random.Random(20261005)
    project = []
    fixed = [(0.3, 0.2), (0.35, -0.24), (-0.37, 0.24), (0.0, 0.0), (0.4, 0.4), (-0.4, -0.4)]
    for pi in range(24):
        dist = dist_readme if pi % 4 else [0.0] * 5
        if pi < len(fixed):
            xn, yn = fixed[pi]
            dist = dist_readme
        else:
            xn, yn = rng2.uniform(-0.4, 0.4), rng2.uniform(-0.4, 0.4)
        Z = rng2.uniform(0.5, 6.0)
        pc = [xn * Z, yn * Z, Z]
        uv = project_camera_model([mpf(c) for c in intr], [mpf(c) for c in dist], [mpf(c) for c in pc])
        fun = corner_residual([mpf(c) for c in intr], [mpf(c) for c in dist],
                              [mpf(1), mpf(0), mpf(0), mpf(0), mpf(0), mpf(0), mpf(0)],
                              [mpf(1), mpf(0), mpf(0), mpf(0)] + [mpf(c) for c in pc],
                              [mpf(0), mpf(0), mpf(0)], [mpf(0), mpf(0)])
        project.append({"intr": intr, "dist": dist, "point_cam": pc, "uv": [float(v) for v in uv],
                        "uv_functor_formula": [float(v) for v in fun]})
    # Point-landmark functor (OpenCVReprojectionError): residual and tangent Jacobians w.r.t. the camera
    # (translation, then half-angle rotation through Plus) and the 3-D point; case 2 and 3 carry a camera quaternion
    # that is NOT unit (UnitQuaternionRotatePoint does not normalise: the result scales with |q|^2)
    rng3 = random.Random(20261006)
    point_cases = []
    for pi in range(8):
        dist = dist_readme if pi % 2 else [0.0] * 5
        qs = rand_quat(rng3, small=True)
        qx = [0.0, 1.0, 0.0, 0.0]
        cam_q = [qs[0] * qx[0] - qs[1] * qx[1] - qs[2] * qx[2] - qs[3] * qx[3],
                 qs[0] * qx[1] + qs[1] * qx[0] + qs[2] * qx[3] - qs[3] * qx[2],
                 qs[0] * qx[2] - qs[1] * qx[3] + qs[2] * qx[0] + qs[3] * qx[1],
                 qs[0] * qx[3] + qs[1] * qx[2] - qs[2] * qx[1] + qs[3] * qx[0]]
        if pi in (2, 3):
            cam_q = [c * 1.05 for c in cam_q]
        cam = cam_q + [rng3.uniform(-0.5, 0.5), rng3.uniform(-0.4, 0.4), rng3.uniform(2.5, 4.0)]
        pt = [rng3.uniform(-1, 1), rng3.uniform(-0.6, 0.6), rng3.gauss(0, 0.05)]
        camm, ptm = [mpf(c) for c in cam], [mpf(c) for c in pt]
        intrm, distm = [mpf(c) for c in intr], [mpf(c) for c in dist]
        r0 = point_residual(intrm, distm, camm, ptm, [mpf(0), mpf(0)])
        uv = [float(r0[0]) + rng3.gauss(0, 1.5), float(r0[1]) + rng3.gauss(0, 1.5)]
        uvm = [mpf(c) for c in uv]
        res = point_residual(intrm, distm, camm, ptm, uvm)
        Jc, Jp = [], []
        for comp in range(2):
            rowc, rowp = [], []
            for a in range(6):
                def fc(e, a=a, comp=comp):
                    d = [mpf(0)] * 6
                    d[a] = e
                    return point_residual(intrm, distm, plus(camm, d), ptm, uvm)[comp]
                rowc.append(diff(fc, mpf(0)))
            for a in range(3):
                def fp(e, a=a, comp=comp):
                    q = list(ptm)
                    q[a] = q[a] + e
                    return point_residual(intrm, distm, camm, q, uvm)[comp]
                rowp.append(diff(fp, mpf(0)))
            Jc.append(rowc)
            Jp.append(rowp)
        point_cases.append({"intr": intr, "dist": dist, "cam_qt": cam, "point": pt, "uv": uv,
                            "residual": [float(v) for v in res],
                            "J_cam": [[float(v) for v in row] for row in Jc],
                            "J_point": [[float(v) for v in row] for row in Jp]})
    out = {"generator": "tests/golden/make_kats.py (mpmath %d digits)" % mp.dps,
           "obs": cases, "plus": plus_cases, "huber": huber, "project_point": project, "point_obs": point_cases}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_residual.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(cases), "observation cases")


if __name__ == "__main__":
    main()
