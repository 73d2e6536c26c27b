"""Deterministic synthetic scenes via tools/libvmm_scene.so.

Stands in for the detector output (marker_detections.json) and the PnP initial guess the
reference computes with OpenCV (/root/reference/src/TagReconstructor.cpp:156,167-230); neither can
run in this image.  Bench/test input only -- not part of the hot path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "tools", "libvmm_scene.so")
_LIB = None


class SceneCfg(C.Structure):
    _fields_ = [("n_cams", C.c_int), ("n_tags", C.c_int), ("seed", C.c_uint64),
                ("visibility", C.c_double), ("noise_px", C.c_double), ("outlier_frac", C.c_double),
                ("outlier_px", C.c_double), ("use_distortion", C.c_int), ("cam_rot_deg", C.c_double),
                ("cam_trans_m", C.c_double), ("tag_rot_deg", C.c_double), ("tag_trans_m", C.c_double),
                ("neighbors_min", C.c_int), ("neighbors_max", C.c_int), ("wall_rows", C.c_int)]


def build(force=False):
    src = os.path.join(_ROOT, "tools", "scene_gen.c")
    if force or not os.path.exists(_SO) or (os.path.exists(src)
                                            and os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "tools"), "-s", "all"])
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        _LIB = C.CDLL(_SO)
        _LIB.vmm_scene_generate.restype = C.c_int
    return _LIB


class SyntheticScene:
    """Arrays of one generated scene.  Poses are rows of (qw,qx,qy,qz,tx,ty,tz)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def n_obs(self):
        return len(self.obs_cam)


def make_scene(config=2, **overrides):
    """config = index into BASELINE.json configs (1-based); overrides are SceneCfg fields."""
    L = _lib()
    cfg = SceneCfg()
    L.vmm_scene_default_cfg(C.byref(cfg), C.c_int(config))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    nc, nt = cfg.n_cams, cfg.n_tags
    cap = nc * nt
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    intr, dist = np.zeros(4), np.zeros(5)
    cam_gt, tag_gt = np.zeros((nc, 7)), np.zeros((nt, 7))
    cam_init, tag_init = np.zeros((nc, 7)), np.zeros((nt, 7))
    tag_wh = np.zeros((nt, 2))
    obs_cam, obs_tag = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    obs_px = np.zeros((cap, 8))
    n = L.vmm_scene_generate(C.byref(cfg), dp(intr), dp(dist), dp(cam_gt), dp(tag_gt), dp(tag_wh),
                             dp(cam_init), dp(tag_init), ip(obs_cam), ip(obs_tag), dp(obs_px),
                             C.c_int(cap))
    if n < 0:
        raise RuntimeError("scene buffer too small")
    return SyntheticScene(config=config, intr=intr, dist=dist, cam_gt=cam_gt, tag_gt=tag_gt,
                          cam_init=cam_init, tag_init=tag_init, tag_wh=tag_wh, fixed_tag=0,
                          obs_cam=obs_cam[:n].copy(), obs_tag=obs_tag[:n].copy(),
                          obs_px=obs_px[:n].copy(), noise_px=cfg.noise_px,
                          robustify=bool(cfg.outlier_frac > 0), visibility=cfg.visibility)


def write_project(scene, directory):
    """Writes the scene as a project directory of the reference's mapping step (SURVEY.md 8(d) "emitted
    files"): camera_intrinsics.json and marker_detections.json in the reference's formats (Appendix B),
    plus ground_truth.json and initial_state.json (reconstruction.json layout) for tests."""
    from . import io as _io
    from .tag_reconstructor import Camera, CameraModel, ReconstructedTag, detection_result_from_arrays
    os.makedirs(directory, exist_ok=True)
    model = CameraModel(*[float(v) for v in scene.intr], distortionCoefficients=scene.dist,
                        verticalResolution=4000, horizontalResolution=6000)
    _io.writeCameraModel(model, os.path.join(directory, "camera_intrinsics.json"))
    det = detection_result_from_arrays(scene.obs_cam, scene.obs_tag, scene.obs_px, scene.tag_wh, len(scene.cam_gt))
    _io.writeDetectionResult(det, os.path.join(directory, "marker_detections.json"))
    for name, cams, tags in (("ground_truth.json", scene.cam_gt, scene.tag_gt),
                             ("initial_state.json", scene.cam_init, scene.tag_init)):
        rt = {t: ReconstructedTag(t, "apriltag_36h11", tags[t, :4], tags[t, 4:], scene.tag_wh[t, 0], scene.tag_wh[t, 1])
              for t in range(len(tags))}
        rc = {c: Camera(c, cams[c, :4], cams[c, 4:]) for c in range(len(cams))}
        _io.exportReconstructions(os.path.join(directory, name), rt, rc, model)
    return model, det
