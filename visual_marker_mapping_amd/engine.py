"""BundleAdjuster: Python handle on the device-resident bundle-adjustment engine (libvmm_ba.so).

This is the flat-array layer under TagReconstructor.doBundleAdjustment; it mirrors what
/root/reference/src/TagReconstructor.cpp:646-743 builds (a ceres::Problem) and solves.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (CONVERGENCE, ELIM_AUTO, ELIM_CAMERAS, ELIM_TAGS, FAILURE, LANDMARK_POINTS,  # noqa: F401
                   LANDMARK_TAG_POSES, NO_CONVERGENCE, PRECISION_F32_ACCUM, PRECISION_F64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_options(**kw):
    o = _lib.Options()
    _lib.lib().vmm_ba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError("unknown solver option %r" % k)
        setattr(o, k, v)
    return o


class BundleAdjuster:
    def __init__(self, intr, dist, cam_qt, tag_qt, tag_wh, fixed_tag, obs_cam, obs_tag, obs_px,
                 device=0, elimination=ELIM_AUTO, rank=0, world_size=1, precision=PRECISION_F64,
                 landmarks=LANDMARK_TAG_POSES, structure_obs=None):
        L = _lib.lib()
        self._h = C.c_void_p()
        self.intr = np.ascontiguousarray(intr, np.float64).reshape(4)
        self.dist = np.ascontiguousarray(dist, np.float64).reshape(5)
        cam_qt = np.ascontiguousarray(cam_qt, np.float64).reshape(-1, 7)
        tag_qt = np.ascontiguousarray(tag_qt, np.float64).reshape(-1, 7)
        tag_wh = np.ascontiguousarray(tag_wh, np.float64).reshape(-1, 2)
        obs_cam = np.ascontiguousarray(obs_cam, np.int32).reshape(-1)
        obs_tag = np.ascontiguousarray(obs_tag, np.int32).reshape(-1)
        obs_px = np.ascontiguousarray(obs_px, np.float64).reshape(-1, 8)
        if not (len(obs_cam) == len(obs_tag) == len(obs_px)):
            raise ValueError("observation arrays differ in length")
        if len(tag_wh) != len(tag_qt):
            raise ValueError("tag_wh and tag_qt differ in length")
        self.n_cams, self.n_tags, self.n_obs = len(cam_qt), len(tag_qt), len(obs_cam)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        p = _lib.Problem()
        p.intr[:] = list(self.intr)
        p.dist[:] = list(self.dist)
        p.n_cams, p.n_tags = self.n_cams, self.n_tags
        p.cam_qt, p.tag_qt, p.tag_wh = dp(cam_qt), dp(tag_qt), dp(tag_wh)
        p.fixed_tag = int(fixed_tag)
        p.n_obs, p.obs_cam, p.obs_tag, p.obs_px = self.n_obs, ip(obs_cam), ip(obs_tag), dp(obs_px)
        co = _lib.CreateOptions()
        L.vmm_ba_default_create_options(C.byref(co))
        co.device, co.elimination, co.rank, co.world_size = device, elimination, rank, world_size
        co.precision = int(precision)
        co.landmarks = int(landmarks)
        self.landmarks = int(landmarks)
        if structure_obs is not None:
            # world_size > 1: the (camera, tag) pairs of ALL ranks' observations, the same on every rank, so that every
            # rank orders the kept family identically (vmm_ba_create_options.structure_obs_*)
            s_cam = np.ascontiguousarray(structure_obs[0], np.int32).reshape(-1)
            s_tag = np.ascontiguousarray(structure_obs[1], np.int32).reshape(-1)
            if len(s_cam) != len(s_tag):
                raise ValueError("structure_obs arrays differ in length")
            co.n_structure_obs, co.structure_obs_cam, co.structure_obs_tag = len(s_cam), ip(s_cam), ip(s_tag)
        _lib.check(L.vmm_ba_create(C.byref(p), C.byref(co), C.byref(self._h)))
        self._allreduce_cb = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().vmm_ba_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- state ----
    def set_state(self, cam_qt=None, tag_qt=None):
        cam = None if cam_qt is None else np.ascontiguousarray(cam_qt, np.float64).reshape(self.n_cams, 7)
        tag = None if tag_qt is None else np.ascontiguousarray(tag_qt, np.float64).reshape(self.n_tags, 7)
        _lib.check(_lib.lib().vmm_ba_set_state(self._h, _ptr(cam), _ptr(tag)))

    def set_observation_mask(self, mask=None):
        """Switches observations off/on (caller's order; None = all on) without rebuilding the handle."""
        if mask is None:
            _lib.check(_lib.lib().vmm_ba_set_observation_mask(self._h, None))
            return
        m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8).reshape(-1)
        if len(m) != self.n_obs:
            raise ValueError("mask length differs from the number of observations")
        _lib.check(_lib.lib().vmm_ba_set_observation_mask(self._h, m.ctypes.data_as(C.c_void_p)))

    def get_state(self):
        cam, tag = np.zeros((self.n_cams, 7)), np.zeros((self.n_tags, 7))
        _lib.check(_lib.lib().vmm_ba_get_state(self._h, _ptr(cam), _ptr(tag)))
        return cam, tag

    def get_points(self):
        """LANDMARK_POINTS handles: (n_tags, 4, 3) world corners LL, LR, UR, UL -- the parameter blocks of
        doBundleAdjustment_points (/root/reference/src/TagReconstructor.cpp:485-492)."""
        pts = np.zeros((self.n_tags, 4, 3))
        _lib.check(_lib.lib().vmm_ba_get_points(self._h, _ptr(pts)))
        return pts

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, hip_stream:int) -> None; sum-all-reduce in place."""
        def tramp(_user, buf, count, stream):
            try:
                fn(int(buf), int(count), int(stream or 0))
                return 0
            except Exception:  # surfaced as VMM_BA_ERR_COLLECTIVE by the library
                import traceback
                traceback.print_exc()
                return 1
        self._allreduce_cb = _lib.ALLREDUCE_FN(tramp)
        _lib.check(_lib.lib().vmm_ba_set_allreduce(self._h, self._allreduce_cb, None))

    def enable_rccl(self, unique_id):
        """Native collective path: ncclAllReduce issued by the library on its own stream, recorded into the LM
        iteration's hipGraph.  `unique_id` = the 128 bytes rank 0 drew with rccl_unique_id(), identical on every
        rank; collective call (every rank of the world must make it)."""
        buf = bytes(unique_id)
        if len(buf) != _lib.RCCL_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.RCCL_ID_BYTES)
        _lib.check(_lib.lib().vmm_ba_enable_rccl(self._h, C.c_char_p(buf)))

    # ---- the hot path ----
    def solve(self, options=None, trace_capacity=0, **kw):
        o = options or default_options(**kw)
        s = _lib.Summary()
        buf = None
        if trace_capacity > 0:
            buf = (_lib.Iteration * trace_capacity)()
            s.trace, s.trace_capacity = buf, trace_capacity
        _lib.check(_lib.lib().vmm_ba_solve(self._h, C.byref(o), C.byref(s)))
        out = {k: getattr(s, k) for k, _ in _lib.Summary._fields_
               if k not in ("trace", "trace_capacity", "reserved")}
        trace = []
        if buf is not None:
            for i in range(min(s.iterations, trace_capacity)):
                trace.append({k: getattr(buf[i], k) for k, _ in _lib.Iteration._fields_ if k != "reserved"})
        out["trace"] = trace
        return out

    def cost(self, robustify=True, huber_a=1.0):
        c = C.c_double(0)
        _lib.check(_lib.lib().vmm_ba_cost(self._h, int(bool(robustify)), float(huber_a), C.byref(c)))
        return c.value

    def reprojection_stats(self, per_corner=True):
        pc, pt = np.zeros(self.n_cams), np.zeros(self.n_tags)
        avg = C.c_double(0)
        corner = np.zeros((self.n_obs, 8)) if per_corner else None
        _lib.check(_lib.lib().vmm_ba_reprojection_stats(self._h, _ptr(pc), _ptr(pt), C.byref(avg), _ptr(corner)))
        return pc, pt, avg.value, corner

    def tag_translation_covariance(self, robustify=False, huber_a=1.0):
        """(n_tags, 3, 3) covariance blocks of the tag translations at the current state -- the
        ceres::Covariance block of /root/reference/src/TagReconstructor.cpp:744-783."""
        cov = np.zeros((self.n_tags, 3, 3))
        _lib.check(_lib.lib().vmm_ba_tag_translation_covariance(self._h, int(bool(robustify)), float(huber_a),
                                                               _ptr(cov)))
        return cov

    # ---- diagnostics ----
    def eval_blocks(self, robustify=True, huber_a=1.0, want_W=True):
        V, U = np.zeros((self.n_cams, 6, 6)), np.zeros((self.n_tags, 6, 6))
        W = np.zeros((self.n_obs, 6, 6)) if want_W else None
        gc, gt = np.zeros((self.n_cams, 6)), np.zeros((self.n_tags, 6))
        if self.landmarks == LANDMARK_POINTS:   # the landmark-side blocks live in the library's point-pair index space
            U = W = gt = None
        c = C.c_double(0)
        _lib.check(_lib.lib().vmm_ba_eval_blocks(self._h, int(bool(robustify)), float(huber_a), C.byref(c),
                                                 _ptr(V), _ptr(U), _ptr(W), _ptr(gc), _ptr(gt)))
        return {"cost": c.value, "V": V, "U": U, "W": W, "g_cam": gc, "g_tag": gt}

    def debug_overlap(self, reps=10):
        """ms of: rank-k update + sum alone, factorisation + solves alone, both back to back, both at once on two streams."""
        out = np.zeros(8)
        _lib.check(_lib.lib().vmm_ba_debug_overlap(self._h, int(reps), _ptr(out)))
        return dict(zip(("syrk_ms", "cholesky_ms", "sequential_ms", "concurrent_ms", "cholesky_in_concurrent_ms",
                         "syrk_in_concurrent_ms", "concurrent_syrk_first_ms", "cholesky_in_concurrent_syrk_first_ms"), out.tolist()))

    def time_kernels(self, options=None, reps=5):
        o = options or default_options()
        t = _lib.KernelTimes()
        _lib.check(_lib.lib().vmm_ba_time_kernels(self._h, C.byref(o), int(reps), C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.KernelTimes._fields_}


def rccl_available():
    """True when librccl.so was resolved in this process (vmm_ba_rccl_available; no device call)."""
    return bool(_lib.lib().vmm_ba_rccl_available())


def rccl_unique_id():
    """128 bytes identifying a new RCCL communicator (ncclGetUniqueId); drawn on rank 0, handed to every rank."""
    buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
    _lib.check(_lib.lib().vmm_ba_rccl_unique_id(buf))
    return buf.raw


def project_points(intr, dist, points_cam, device=0):
    """CameraModel::projectPoint (/root/reference/src/CameraModel.cpp:6-26) for an (n,3) batch."""
    intr = np.ascontiguousarray(intr, np.float64).reshape(4)
    dist = np.ascontiguousarray(dist, np.float64).reshape(5)
    pts = np.ascontiguousarray(points_cam, np.float64).reshape(-1, 3)
    uv = np.zeros((len(pts), 2))
    _lib.check(_lib.lib().vmm_ba_project_points(_ptr(intr), _ptr(dist), len(pts), _ptr(pts), _ptr(uv), device))
    return uv


def pose_plus(qt, delta, device=0):
    """Plus(qt, delta) for (n,7) poses and (n,6) tangent steps with the engine's device function
    (Ceres QuaternionParameterization::Plus on q, addition on t; tangent = translation then rotation)."""
    qt = np.ascontiguousarray(qt, np.float64).reshape(-1, 7)
    delta = np.ascontiguousarray(delta, np.float64).reshape(-1, 6)
    out = np.zeros_like(qt)
    _lib.check(_lib.lib().vmm_ba_pose_plus(len(qt), _ptr(qt), _ptr(delta), _ptr(out), device))
    return out


def dense_spd_solve(A, b, device=0):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(len(b))
    info = C.c_int(0)
    _lib.check(_lib.lib().vmm_ba_dense_spd_solve(device, len(b), _ptr(A), _ptr(b), _ptr(x), C.byref(info)))
    return x, info.value


def dense_syrk(Z, device=0):
    Z = np.ascontiguousarray(Z, np.float64)
    k, n = Z.shape
    Cm = np.zeros((n, n))
    _lib.check(_lib.lib().vmm_ba_dense_syrk(device, k, n, _ptr(Z), _ptr(Cm)))
    return Cm
