"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (visual_marker_mapping_amd) never imports this module.
PARITY UNPINNED against Ceres -- see oracle/vmm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DENSE_NORMAL, SCHUR_ELIM_TAGS, SCHUR_ELIM_CAMS, SCHUR_AUTO = 0, 1, 2, 3
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2


class Problem(C.Structure):
    _fields_ = [("intr", C.c_double * 4), ("dist", C.c_double * 5), ("n_cams", C.c_int),
                ("cam_qt", C.POINTER(C.c_double)), ("n_tags", C.c_int),
                ("tag_qt", C.POINTER(C.c_double)), ("tag_wh", C.POINTER(C.c_double)),
                ("fixed_tag", C.c_int), ("n_obs", C.c_int), ("obs_cam", C.POINTER(C.c_int)),
                ("obs_tag", C.POINTER(C.c_int)), ("obs_px", C.POINTER(C.c_double)),
                ("landmark_points", C.c_int), ("fixed_tag2", C.c_int)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("robustify", C.c_int), ("huber_a", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("max_num_consecutive_invalid_steps", C.c_int),
                ("jacobi_scaling", C.c_int), ("linear_solver", C.c_int), ("num_threads", C.c_int)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("iterations", C.c_int),
                ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("num_jacobian_evals", C.c_int), ("num_cost_evals", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("time_total_s", C.c_double), ("time_eval_s", C.c_double),
                ("time_linear_s", C.c_double), ("trace", C.POINTER(Iteration)),
                ("trace_capacity", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "vmm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.vo_cost.restype = C.c_double
        L.vo_solve.restype = C.c_int
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def default_options(**kw):
    o = Options()
    lib().vo_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Scene:
    """Flat problem arrays (owned numpy) + the ctypes view handed to the oracle."""

    def __init__(self, intr, dist, cam_qt, tag_qt, tag_wh, fixed_tag, obs_cam, obs_tag, obs_px,
                 landmark_points=False, fixed_tag2=-1):
        self.landmark_points, self.fixed_tag2 = bool(landmark_points), int(fixed_tag2)
        self.intr = np.ascontiguousarray(intr, np.float64).reshape(4)
        self.dist = np.ascontiguousarray(dist, np.float64).reshape(5)
        self.cam_qt = np.array(cam_qt, np.float64).reshape(-1, 7).copy()
        self.tag_qt = np.array(tag_qt, np.float64).reshape(-1, 7).copy()
        self.tag_wh = np.ascontiguousarray(tag_wh, np.float64).reshape(-1, 2)
        self.fixed_tag = int(fixed_tag)
        self.obs_cam = np.ascontiguousarray(obs_cam, np.int32).reshape(-1)
        self.obs_tag = np.ascontiguousarray(obs_tag, np.int32).reshape(-1)
        self.obs_px = np.ascontiguousarray(obs_px, np.float64).reshape(-1, 8)

    def c_problem(self):
        p = Problem()
        p.intr[:] = list(self.intr)
        p.dist[:] = list(self.dist)
        p.n_cams, p.cam_qt = len(self.cam_qt), _dp(self.cam_qt)
        p.n_tags, p.tag_qt, p.tag_wh = len(self.tag_qt), _dp(self.tag_qt), _dp(self.tag_wh)
        p.fixed_tag = self.fixed_tag
        p.landmark_points, p.fixed_tag2 = int(self.landmark_points), self.fixed_tag2
        p.n_obs, p.obs_cam, p.obs_tag, p.obs_px = (len(self.obs_cam), _ip(self.obs_cam),
                                                   _ip(self.obs_tag), _dp(self.obs_px))
        return p


def cost(scene, opts=None):
    opts = opts or default_options()
    p = scene.c_problem()
    return lib().vo_cost(C.byref(p), C.byref(opts))


def solve(scene, opts=None, trace_capacity=2048):
    """Runs vo_solve in place on scene.cam_qt / scene.tag_qt.  Returns (summary dict, trace list)."""
    opts = opts or default_options()
    p = scene.c_problem()
    s = Summary()
    buf = (Iteration * trace_capacity)()
    s.trace, s.trace_capacity = buf, trace_capacity
    lib().vo_solve(C.byref(p), C.byref(opts), C.byref(s))
    out = {k: getattr(s, k) for k, _ in Summary._fields_ if k not in ("trace", "trace_capacity")}
    n = min(s.iterations, trace_capacity)
    trace = [{k: getattr(buf[i], k) for k, _ in Iteration._fields_} for i in range(n)]
    return out, trace


def obs_eval(intr, dist, cam_qt, tag_qt, wh, px, jac=True):
    intr, dist = np.ascontiguousarray(intr, np.float64), np.ascontiguousarray(dist, np.float64)
    cam_qt, tag_qt = np.ascontiguousarray(cam_qt, np.float64), np.ascontiguousarray(tag_qt, np.float64)
    wh, px = np.ascontiguousarray(wh, np.float64), np.ascontiguousarray(px, np.float64)
    r = np.zeros(8)
    Jc, Jt = np.zeros((8, 6)), np.zeros((8, 6))
    lib().vo_obs_eval(_dp(intr), _dp(dist), _dp(cam_qt), _dp(tag_qt), _dp(wh), _dp(px), _dp(r),
                      _dp(Jc) if jac else None, _dp(Jt) if jac else None)
    return (r, Jc, Jt) if jac else r


def point_eval(intr, dist, cam_qt, point, uv, jac=True):
    """OpenCVReprojectionError (TagReconstructionCostFunction.h:21-68): residual (2,), Jc (2,6), Jp (2,3)."""
    intr, dist = np.ascontiguousarray(intr, np.float64), np.ascontiguousarray(dist, np.float64)
    cam_qt, point = np.ascontiguousarray(cam_qt, np.float64), np.ascontiguousarray(point, np.float64)
    uv = np.ascontiguousarray(uv, np.float64)
    r, Jc, Jp = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3))
    lib().vo_point_eval(_dp(intr), _dp(dist), _dp(cam_qt), _dp(point), _dp(uv), _dp(r), _dp(Jc) if jac else None,
                        _dp(Jp) if jac else None)
    return (r, Jc, Jp) if jac else r


def point_scene(intr, dist, cam_qt, tag_qt, tag_wh, fixed_tag, obs_cam, obs_tag, obs_px):
    """The point-landmark problem of doBundleAdjustment_points (src/TagReconstructor.cpp:457-644) for the same
    detections: every tag becomes its four world corners (computeMarkerCorners3D, :483: Eigen rotation, no
    normalisation), stored as two point pairs; every tag observation becomes two pair observations.
    Returns (Scene, points) with points (n_tags, 4, 3) the initial corners."""
    tag_qt = np.asarray(tag_qt, np.float64).reshape(-1, 7)
    tag_wh = np.asarray(tag_wh, np.float64).reshape(-1, 2)
    w, x, y, z = tag_qt[:, 0], tag_qt[:, 1], tag_qt[:, 2], tag_qt[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    pts = np.zeros((len(tag_qt), 4, 3))
    for k, (sx, sy) in enumerate(((-1, -1), (1, -1), (1, 1), (-1, 1))):
        loc = np.stack([sx * tag_wh[:, 0] / 2, sy * tag_wh[:, 1] / 2, np.zeros(len(tag_wh))], axis=1)
        pts[:, k] = np.einsum("nij,nj->ni", R, loc) + tag_qt[:, 4:]
    pair = np.zeros((2 * len(tag_qt), 7))
    pair[0::2, :6] = pts[:, 0:2].reshape(-1, 6)
    pair[1::2, :6] = pts[:, 2:4].reshape(-1, 6)
    obs_cam, obs_tag = np.asarray(obs_cam, np.int32), np.asarray(obs_tag, np.int32)
    px = np.asarray(obs_px, np.float64).reshape(-1, 8)
    oc = np.repeat(obs_cam, 2)
    ot = np.stack([2 * obs_tag, 2 * obs_tag + 1], axis=1).reshape(-1)
    opx = np.zeros((2 * len(px), 8))
    opx[0::2, :4] = px[:, :4]
    opx[1::2, :4] = px[:, 4:]
    sc = Scene(intr, dist, cam_qt, pair, np.ones((len(pair), 2)), 2 * fixed_tag if fixed_tag >= 0 else -1, oc, ot, opx,
               landmark_points=True, fixed_tag2=2 * fixed_tag + 1 if fixed_tag >= 0 else -1)
    return sc, pts


def scene_points(scene):
    """(n_tags, 4, 3) points of a point_scene after a solve."""
    return scene.tag_qt[:, :6].reshape(-1, 2, 2, 3).reshape(-1, 4, 3)


def huber(a, s):
    rho = np.zeros(3)
    lib().vo_huber(C.c_double(a), C.c_double(s), _dp(rho))
    return rho


def pose_plus(qt, delta):
    qt, delta = np.ascontiguousarray(qt, np.float64), np.ascontiguousarray(delta, np.float64)
    out = np.zeros(7)
    lib().vo_pose_plus(_dp(qt), _dp(delta), _dp(out))
    return out


def project_point(intr, dist, pc):
    intr, dist, pc = (np.ascontiguousarray(a, np.float64) for a in (intr, dist, pc))
    uv = np.zeros(2)
    lib().vo_project_point(_dp(intr), _dp(dist), _dp(pc), _dp(uv))
    return uv


def reprojection_stats(scene):
    p = scene.c_problem()
    pc, pt = np.zeros(len(scene.cam_qt)), np.zeros(len(scene.tag_qt))
    avg = C.c_double(0)
    corner = np.zeros((len(scene.obs_cam), 8))
    lib().vo_reprojection_stats(C.byref(p), _dp(pc), _dp(pt), C.byref(avg), _dp(corner))
    return pc, pt, avg.value, corner


def tag_translation_covariance(scene, opts=None):
    """(n_tags, 3, 3) covariance blocks of the tag translations; raises if J^T J is singular."""
    opts = opts or default_options()
    p = scene.c_problem()
    cov = np.zeros((len(scene.tag_qt), 3, 3))
    if lib().vo_tag_translation_covariance(C.byref(p), C.byref(opts), _dp(cov)) != 0:
        raise RuntimeError("J^T J is not positive definite")
    return cov

