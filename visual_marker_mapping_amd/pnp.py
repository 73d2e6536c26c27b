"""Pose initialisation from 2-D/3-D correspondences for the incremental driver (host side, numpy).

The reference initialises every new camera with cv::solvePnPRansac and every new tag with
cv::solvePnP(CV_ITERATIVE) (/root/reference/src/EigenCVConversions.cpp:38-106, called from
src/TagReconstructor.cpp:156,167-230,291-310).  OpenCV is not installed here, so this
module restates the published method of OpenCV's iterative PnP -- planar point sets: homography of the
undistorted normalised points, decomposition, orthonormalisation; other sets: 3x4 DLT; then
Levenberg-Marquardt on the pixel reprojection error over (Rodrigues vector, translation) -- and a plain
RANSAC loop with OpenCV's default parameters (100 iterations, 8 px, confidence 0.99) around it.

These are INITIAL GUESSES for the bundle adjustment that follows each of them; they are not on the
measured hot path and their parity with OpenCV is UNPINNED (no OpenCV outputs exist in the reference tree).
What tests pin instead: exact recovery on noise-free data (planar and non-planar), and that the GPU bundle
adjustment started from them reaches the optimum it reaches from the ground-truth-perturbed start.
"""
import numpy as np


def rodrigues(r):
    """Rotation vector -> rotation matrix (cv::Rodrigues)."""
    r = np.asarray(r, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-300:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(th) * Kx + (1.0 - np.cos(th)) * (Kx @ Kx)


def rodrigues_inv(R):
    """Rotation matrix -> rotation vector."""
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = float(np.arccos(c))
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-12:
        return 0.5 * w
    if np.pi - th < 1e-6:   # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        i = int(np.argmax(ax))
        ax = A[:, i] / ax[i]
        ax /= np.linalg.norm(ax)
        if np.dot(ax, w) < 0:
            ax = -ax
        return th * ax
    return th / (2.0 * np.sin(th)) * w


def quat_from_R(R):
    """Eigen::Quaterniond(R) as (w, x, y, z), w >= 0 branch as Eigen picks it (trace test)."""
    t = np.trace(R)
    if t > 0.0:
        s = np.sqrt(t + 1.0) * 2.0
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2.0
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def project(X, R, t, intr, dist):
    """Pixel projection of world points X (n,3) with OpenCV's distortion model (cv::projectPoints, the model
    solvePnP minimises; /root/reference/src/EigenCVConversions.cpp:38-106 calls OpenCV).  Deliberately NOT
    CameraModel::projectPoint, whose y tangential term uses the already distorted x (src/CameraModel.cpp:20-23)."""
    fx, fy, cx, cy = intr
    k1, k2, p1, p2, k3 = dist
    P = X @ R.T + t
    x = P[:, 0] / P[:, 2]
    y = P[:, 1] / P[:, 2]
    r2 = x * x + y * y
    rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], axis=1)


def undistort_normalized(px, intr, dist, iters=20):
    """Pixels -> undistorted normalised image coordinates (fixed-point iteration as cv::undistortPoints)."""
    fx, fy, cx, cy = intr
    k1, k2, p1, p2, k3 = dist
    px = np.asarray(px, np.float64).reshape(-1, 2)
    x0 = (px[:, 0] - cx) / fx
    y0 = (px[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x = (x0 - dx) / rad
        y = (y0 - dy) / rad
    return np.stack([x, y], axis=1)


def _normalise2d(p):
    c = p.mean(axis=0)
    d = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
    s = np.sqrt(2.0) / d if d > 0 else 1.0
    T = np.array([[s, 0.0, -s * c[0]], [0.0, s, -s * c[1]], [0.0, 0.0, 1.0]])
    return (p - c) * s, T


def homography_dlt(src, dst):
    """H with dst ~ H src (normalised DLT, least squares over all points)."""
    a, Ta = _normalise2d(np.asarray(src, np.float64))
    b, Tb = _normalise2d(np.asarray(dst, np.float64))
    n = a.shape[0]
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2] = a
    A[0::2, 2] = 1.0
    A[0::2, 6:8] = -b[:, 0:1] * a
    A[0::2, 8] = -b[:, 0]
    A[1::2, 3:5] = a
    A[1::2, 5] = 1.0
    A[1::2, 6:8] = -b[:, 1:2] * a
    A[1::2, 8] = -b[:, 1]
    _, _, Vt = np.linalg.svd(A)
    H = Vt[-1].reshape(3, 3)
    return np.linalg.inv(Tb) @ H @ Ta


def _orthonormalise(R):
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = U @ np.diag([1.0, 1.0, -1.0]) @ Vt
    return R


def _pose_planar(Xp, xn):
    """Pose of the plane z = 0 (points Xp (n,2)) from normalised image points."""
    H = homography_dlt(Xp, xn)
    h1, h2, h3 = H[:, 0], H[:, 1], H[:, 2]
    n1, n2 = np.linalg.norm(h1), np.linalg.norm(h2)
    s = 2.0 / (n1 + n2)
    if h3[2] * s < 0:   # the plane is in front of the camera
        s = -s
    r1, r2 = h1 / n1 * np.sign(s), h2 / n2 * np.sign(s)
    R = _orthonormalise(np.stack([r1, r2, np.cross(r1, r2)], axis=1))
    return R, h3 * s


def _pose_dlt(X, xn):
    """Pose from the 3x4 DLT of non-coplanar points."""
    n = X.shape[0]
    c = X.mean(axis=0)
    sc = np.sqrt(((X - c) ** 2).sum(axis=1)).mean()
    sc = np.sqrt(3.0) / sc if sc > 0 else 1.0
    Xn = (X - c) * sc
    Xh = np.concatenate([Xn, np.ones((n, 1))], axis=1)
    A = np.zeros((2 * n, 12))
    A[0::2, 0:4] = Xh
    A[0::2, 8:12] = -xn[:, 0:1] * Xh
    A[1::2, 4:8] = Xh
    A[1::2, 8:12] = -xn[:, 1:2] * Xh
    _, _, Vt = np.linalg.svd(A)
    P = Vt[-1].reshape(3, 4)
    M = P[:, :3]
    if np.linalg.det(M) < 0:
        P = -P
        M = P[:, :3]
    s = 1.0 / np.cbrt(max(np.linalg.det(M), 1e-300))
    R = _orthonormalise(M * s)
    tn = P[:, 3] * s                      # sc X_cam = R Xn + tn with Xn = (X - c) sc
    return R, tn / sc - R @ c


def _initial_pose(X, xn):
    """Planarity-aware initial pose (the cvFindExtrinsicCameraParams2 recipe)."""
    c = X.mean(axis=0)
    _, w, Vt = np.linalg.svd(X - c, full_matrices=False)
    planar = X.shape[0] < 6 or w[2] ** 2 < 1e-3 * w[1] ** 2
    if planar:
        Rp = Vt.copy()                    # rows: in-plane axes, normal
        if np.linalg.det(Rp) < 0:
            Rp[2] = -Rp[2]
        Xl = (X - c) @ Rp.T               # plane coordinates (z ~ 0)
        R, t = _pose_planar(Xl[:, :2], xn)
        return R @ Rp, t - R @ Rp @ c     # X_cam = R (Rp (X - c)) + t
    R, t = _pose_dlt(X, xn)
    return R, t


def _rodrigues_batch(r):
    """(m,3) rotation vectors -> (m,3,3) rotation matrices."""
    th = np.linalg.norm(r, axis=1)
    safe = np.where(th > 1e-300, th, 1.0)
    k = r / safe[:, None]
    Kx = np.zeros((r.shape[0], 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2] = -k[:, 2], k[:, 1]
    Kx[:, 1, 0], Kx[:, 1, 2] = k[:, 2], -k[:, 0]
    Kx[:, 2, 0], Kx[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    R = np.eye(3)[None] + s * Kx + (1.0 - c) * (Kx @ Kx)
    R[th <= 1e-300] = np.eye(3)
    return R


def _residuals_batch(P, X, px, intr, dist):
    """Residual vectors (m, 2n) of m poses P = (rvec, t) at once -- the arithmetic of project()."""
    fx, fy, cx, cy = intr
    k1, k2, p1, p2, k3 = dist
    R = _rodrigues_batch(P[:, :3])
    Pc = np.einsum("mij,nj->mni", R, X) + P[:, None, 3:]
    x = Pc[:, :, 0] / Pc[:, :, 2]
    y = Pc[:, :, 1] / Pc[:, :, 2]
    r2 = x * x + y * y
    rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    res = np.stack([fx * xd + cx - px[None, :, 0], fy * yd + cy - px[None, :, 1]], axis=2)
    return res.reshape(P.shape[0], -1)


def refine_pose(R, t, X, px, intr, dist, max_iter=20, eps=1.1920929e-07):
    """Levenberg-Marquardt on the pixel reprojection error over (rvec, t); OpenCV's termination
    (20 iterations or a relative parameter change below FLT_EPSILON).  Central-difference Jacobian, all
    twelve perturbed poses evaluated in one vectorised call."""
    p = np.concatenate([rodrigues_inv(R), np.asarray(t, np.float64)])
    r = _residuals_batch(p[None], X, px, intr, dist)[0]
    cost = float(r @ r)
    lam = 1e-3
    for _ in range(max_iter):
        h = 1e-6 * np.maximum(1.0, np.abs(p))
        Pp = np.concatenate([p[None] + np.diag(h), p[None] - np.diag(h)])
        rr = _residuals_batch(Pp, X, px, intr, dist)
        J = ((rr[:6] - rr[6:]) / (2.0 * h)[:, None]).T
        A = J.T @ J
        g = J.T @ r
        improved = False
        rel = 0.0
        for _ in range(10):
            try:
                step = np.linalg.solve(A + lam * np.diag(np.maximum(np.diag(A), 1e-12)), -g)
            except np.linalg.LinAlgError:
                lam *= 10.0
                continue
            q = p + step
            rq = _residuals_batch(q[None], X, px, intr, dist)[0]
            cq = float(rq @ rq)
            if np.isfinite(cq) and cq < cost:
                rel = np.linalg.norm(step) / max(np.linalg.norm(p), 1e-300)
                p, r, cost = q, rq, cq
                lam = max(lam / 10.0, 1e-12)
                improved = True
                break
            lam *= 10.0
        if not improved or rel < eps:
            break
    return rodrigues(p[:3]), p[3:].copy()


def solvePnP(objectPoints, observations, intr, dist):
    """R, t with x_cam = R X + t (stands in for solvePnPEigen, src/EigenCVConversions.cpp:38-63)."""
    X = np.asarray(objectPoints, np.float64).reshape(-1, 3)
    px = np.asarray(observations, np.float64).reshape(-1, 2)
    if X.shape[0] != px.shape[0] or X.shape[0] < 4:
        raise RuntimeError("solvePnP needs at least 4 correspondences of equal count")
    xn = undistort_normalized(px, intr, dist)
    R, t = _initial_pose(X, xn)
    return refine_pose(R, t, X, px, intr, dist)


def solvePnPRansac(objectPoints, observations, intr, dist, iterations=100, reprojection_error=8.0,
                   confidence=0.99, seed=0):
    """R, t (stands in for solvePnPRansacEigen, src/EigenCVConversions.cpp:65-106; OpenCV defaults).

    The reference narrows the points to float32 before the call; that noise is
    not reproduced."""
    X = np.asarray(objectPoints, np.float64).reshape(-1, 3)
    px = np.asarray(observations, np.float64).reshape(-1, 2)
    n = X.shape[0]
    if n != px.shape[0]:
        raise RuntimeError("For solvePnPRansac the same number of objectPoints and observations is needed. "
                           "Num objectPoints: %d Num observations: %d" % (n, px.shape[0]))
    m = 6   # sample size of the planarity-aware solver
    if n <= m:
        return solvePnP(X, px, intr, dist)
    rng = np.random.default_rng(seed)
    best_inl, best = None, -1
    # all points first: detections are rarely contaminated, and then no sampling is needed
    cand = [np.arange(n)] + [None] * iterations
    max_it = iterations
    it = 0
    while it <= max_it:
        idx = cand[it] if cand[it] is not None else rng.choice(n, size=m, replace=False)
        it += 1
        try:
            R, t = solvePnP(X[idx], px[idx], intr, dist)
        except (np.linalg.LinAlgError, RuntimeError):
            continue
        err = np.linalg.norm(project(X, R, t, intr, dist) - px, axis=1)
        inl = np.isfinite(err) & (err < reprojection_error)
        k = int(inl.sum())
        if k > best:
            best, best_inl = k, inl
            w = k / n
            if w >= 1.0:
                if len(idx) == n:
                    return R, t   # the all-points solve itself: nothing to re-fit
                break
            denom = np.log(max(1.0 - w ** m, 1e-300))
            max_it = min(iterations, int(np.ceil(np.log(1.0 - confidence) / denom))) if denom < 0 else iterations
    if best_inl is None or best < 4:
        return solvePnP(X, px, intr, dist)
    return solvePnP(X[best_inl], px[best_inl], intr, dist)
