"""ctypes binding of libvmm_ba.so (include/vmm_ba.h).

There is no CPU fallback: if the library is missing or no MI355X is visible the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VMM_BA_LIB selects another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("VMM_BA_LIB") or os.path.join(_HERE, "libvmm_ba.so")

ABI_VERSION = 5          # VMM_BA_ABI_VERSION of include/vmm_ba.h
RCCL_ID_BYTES = 128      # VMM_BA_RCCL_ID_BYTES
PRECISION_F64, PRECISION_F32_ACCUM = 0, 1
LANDMARK_TAG_POSES, LANDMARK_POINTS = 0, 1
OK, ERR_ARGUMENT, ERR_HIP, ERR_COLLECTIVE, ERR_STATE, ERR_NUMERIC = 0, 1, 2, 3, 4, 5
ELIM_AUTO, ELIM_TAGS, ELIM_CAMERAS = 0, 1, 2
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2

# every symbol include/vmm_ba.h declares
EXPORTS = ["vmm_ba_last_error", "vmm_ba_abi_version", "vmm_ba_default_options",
           "vmm_ba_default_create_options", "vmm_ba_create", "vmm_ba_destroy", "vmm_ba_set_state",
           "vmm_ba_get_state", "vmm_ba_get_points", "vmm_ba_set_allreduce", "vmm_ba_rccl_available", "vmm_ba_rccl_unique_id",
           "vmm_ba_enable_rccl",
           "vmm_ba_set_observation_mask", "vmm_ba_solve", "vmm_ba_cost",
           "vmm_ba_reprojection_stats", "vmm_ba_tag_translation_covariance", "vmm_ba_project_points", "vmm_ba_eval_blocks",
           "vmm_ba_dense_spd_solve", "vmm_ba_dense_syrk", "vmm_ba_time_kernels", "vmm_ba_pose_plus", "vmm_ba_debug_overlap",
           "vmm_ba_debug_chol_schedule", "vmm_ba_debug_chol_tile"]


class Problem(C.Structure):
    _fields_ = [("intr", C.c_double * 4), ("dist", C.c_double * 5), ("n_cams", C.c_int32),
                ("n_tags", C.c_int32), ("cam_qt", C.POINTER(C.c_double)),
                ("tag_qt", C.POINTER(C.c_double)), ("tag_wh", C.POINTER(C.c_double)),
                ("fixed_tag", C.c_int32), ("n_obs", C.c_int64), ("obs_cam", C.POINTER(C.c_int32)),
                ("obs_tag", C.POINTER(C.c_int32)), ("obs_px", C.POINTER(C.c_double))]


class CreateOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("elimination", C.c_int32), ("rank", C.c_int32),
                ("world_size", C.c_int32), ("precision", C.c_int32), ("landmarks", C.c_int32),
                ("n_structure_obs", C.c_int64), ("structure_obs_cam", C.POINTER(C.c_int32)),
                ("structure_obs_tag", C.POINTER(C.c_int32))]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("robustify", C.c_int32), ("huber_a", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("max_num_consecutive_invalid_steps", C.c_int32),
                ("jacobi_scaling", C.c_int32), ("num_threads", C.c_int32), ("poll_interval", C.c_int32)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32),
                ("step_is_successful", C.c_int32), ("reserved", C.c_int32), ("cost", C.c_double),
                ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
                ("step_norm", C.c_double), ("relative_decrease", C.c_double),
                ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("iterations", C.c_int32),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("num_lm_iterations", C.c_int32), ("num_jacobian_evals", C.c_int32),
                ("num_cost_evals", C.c_int32), ("elimination", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("time_solve_s", C.c_double),
                ("trace", C.POINTER(Iteration)), ("trace_capacity", C.c_int32), ("reserved", C.c_int32),
                ("time_eval_s", C.c_double), ("time_eliminate_s", C.c_double), ("time_factor_solve_s", C.c_double),
                ("time_step_s", C.c_double), ("time_control_s", C.c_double),
                ("num_sync_timeouts", C.c_int32), ("sync_timeout_kernels", C.c_int32),
                ("block_sparse", C.c_int32), ("tree_ordering", C.c_int32)]


class KernelTimes(C.Structure):
    _fields_ = [("eval_elim_ms", C.c_double), ("eval_keep_ms", C.c_double), ("cost_ms", C.c_double),
                ("form_z_ms", C.c_double), ("syrk_ms", C.c_double), ("cholesky_ms", C.c_double),
                ("backsub_ms", C.c_double), ("lm_iteration_ms", C.c_double), ("n_obs", C.c_int64),
                ("reduced_dim", C.c_int32), ("elim_dim", C.c_int32), ("schur_sparse", C.c_int32),
                ("syrk_wide", C.c_int32), ("schur_flops", C.c_double), ("chol_flops", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_LIB = None


class VmmBaError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libvmm_ba status %d: %s" % (status, message))
        self.status = status


def lib():
    """Loads libvmm_ba.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with __graft_entry__.build() or "
                              "make -C visual_marker_mapping_amd/csrc" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.vmm_ba_last_error.restype = C.c_char_p
        L.vmm_ba_abi_version.restype = C.c_int
        if L.vmm_ba_abi_version() != ABI_VERSION:
            raise ImportError("%s has ABI version %d, this binding needs %d: rebuild it"
                              % (LIB_PATH, L.vmm_ba_abi_version(), ABI_VERSION))
        L.vmm_ba_destroy.restype = None
        L.vmm_ba_destroy.argtypes = [C.c_void_p]
        L.vmm_ba_create.argtypes = [C.POINTER(Problem), C.POINTER(CreateOptions), C.POINTER(C.c_void_p)]
        L.vmm_ba_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vmm_ba_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vmm_ba_get_points.argtypes = [C.c_void_p, C.c_void_p]
        L.vmm_ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.vmm_ba_rccl_available.restype = C.c_int
        L.vmm_ba_rccl_available.argtypes = []
        L.vmm_ba_rccl_unique_id.argtypes = [C.c_void_p]
        L.vmm_ba_enable_rccl.argtypes = [C.c_void_p, C.c_void_p]
        L.vmm_ba_set_observation_mask.argtypes = [C.c_void_p, C.c_void_p]
        L.vmm_ba_solve.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Summary)]
        L.vmm_ba_cost.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.vmm_ba_reprojection_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_double), C.c_void_p]
        L.vmm_ba_tag_translation_covariance.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        L.vmm_ba_project_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_int]
        L.vmm_ba_eval_blocks.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vmm_ba_dense_spd_solve.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_int)]
        L.vmm_ba_dense_syrk.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vmm_ba_pose_plus.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vmm_ba_time_kernels.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int,
                                          C.POINTER(KernelTimes)]
        L.vmm_ba_debug_overlap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vmm_ba_debug_chol_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.vmm_ba_debug_chol_tile.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def check(status):
    if status != OK:
        raise VmmBaError(status, lib().vmm_ba_last_error().decode("utf-8", "replace"))
