"""A spin give-up is not an indefinite matrix.

k_chol_dataflow and k_backsolve_chain (visual_marker_mapping_amd/csrc/kernels_chol.hip) hand data between workgroups
inside one launch with bounded spins.  A spin that gives up -- a GPU time-sliced between rank processes, a profiler
serialising workgroups -- used to raise LmCtl::lin_fail, which the trust-region policy (Ceres' HandleInvalidStep:
radius shrink, retry, FAILURE after five) takes for a failed linear solve: a scheduling event silently changed the LM
trajectory.  Now it raises LmCtl::sync_timeout, pauses the pass, and the host redoes that pass's factorisation on the
launch-per-block-column path; vmm_ba_summary reports how often and in which kernel.  The debug variables
VMM_BA_DEBUG_SPIN_LIMIT / _KERNEL / _ONCE (read at vmm_ba_create) force the give-ups here.

Reference semantics at stake: the exact linear solve inside ceres::Solve (src/TagReconstructor.cpp:737-738) -- a
recovered pass must give the same step as an undisturbed one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAR = dict(n_cams=40, n_tags=24, cam_rot_deg=50.0, cam_trans_m=0.8, tag_rot_deg=50.0, tag_trans_m=0.5)


def _solve(s, robust, elim=None):
    from visual_marker_mapping_amd import engine as eng
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px, elimination=eng.ELIM_AUTO if elim is None else elim)
    try:
        out = ba.solve(eng.default_options(robustify=robust), trace_capacity=256)
        cam, tag = ba.get_state()
        again = None
        if out["num_sync_timeouts"]:
            # the handle keeps working: a second solve from the start walks the same path
            ba.set_state(s.cam_init, s.tag_init)
            again = ba.solve(eng.default_options(robustify=robust))
    finally:
        ba.close()
    return out, cam, tag, again


def _same(a, b, ca, ta, cb, tb):
    assert a["termination_type"] == b["termination_type"] and a["iterations"] == b["iterations"]
    assert a["num_unsuccessful_steps"] == b["num_unsuccessful_steps"]
    for x, y in zip(a["trace"], b["trace"]):
        assert x["step_is_valid"] == y["step_is_valid"] and x["step_is_successful"] == y["step_is_successful"]
        np.testing.assert_allclose(x["cost"], y["cost"], rtol=1e-9)
        np.testing.assert_allclose(x["trust_region_radius"], y["trust_region_radius"], rtol=1e-6)
    np.testing.assert_allclose(ca, cb, rtol=0, atol=1e-8 * np.abs(cb).max())
    np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-8 * np.abs(tb).max())


@pytest.mark.parametrize("kernel,bit", [("df", 1), ("chain", 2), ("both", 1)])
@pytest.mark.parametrize("once", ["0", "1"])
@pytest.mark.parametrize("far", [False, True])
def test_forced_spin_give_ups_keep_the_trajectory(monkeypatch, kernel, bit, once, far):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    # 24 / 30 kept poses: reduced order 144 / 180 = 3 blocks, so both one-launch kernels have workgroups that wait
    s = make_scene(1, **FAR) if far else make_scene(5, n_cams=60, n_tags=30)
    robust = 0 if far else 1
    ref, cam0, tag0, _ = _solve(s, robust)
    assert ref["num_sync_timeouts"] == 0 and ref["sync_timeout_kernels"] == 0
    if far:
        assert ref["num_unsuccessful_steps"] >= 2
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_LIMIT", "1")
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_KERNEL", kernel)
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_ONCE", once)
    out, cam, tag, again = _solve(s, robust)
    assert out["termination_type"] == eng.CONVERGENCE
    assert out["num_sync_timeouts"] >= 1 and out["sync_timeout_kernels"] & bit
    if once == "1":
        assert out["num_sync_timeouts"] == 1          # the one-launch kernels are back for the rest of the solve
    else:
        assert out["num_sync_timeouts"] == out["num_lm_iterations"]   # every pass was redone
    _same(out, ref, cam, tag, cam0, tag0)
    assert again["iterations"] == ref["iterations"]
    np.testing.assert_allclose(again["final_cost"], ref["final_cost"], rtol=1e-9)


def test_give_up_and_indefinite_matrix_are_told_apart(monkeypatch):
    """vmm_ba_dense_spd_solve: with the spins forced to give up an SPD system is still solved (fallback path) and an
    indefinite one is still reported through info -- by the fallback's own pivots, not by the time-out."""
    from visual_marker_mapping_amd import engine as eng
    rng = np.random.default_rng(7)
    n = 300
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x_ref = np.linalg.solve(A, b)
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_LIMIT", "1")
    x, info = eng.dense_spd_solve(A, b)
    assert info == 0
    np.testing.assert_allclose(x, x_ref, rtol=0, atol=1e-10 * np.abs(x_ref).max())
    B = A.copy()
    B[200, 200] = -1.0
    _, info = eng.dense_spd_solve(B, b)
    assert info != 0


def test_covariance_survives_a_give_up(monkeypatch):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=60, n_tags=30)

    def cov():
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_gt, s.tag_gt, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
        try:
            return ba.tag_translation_covariance(robustify=False)
        finally:
            ba.close()
    ref = cov()
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_LIMIT", "1")
    got = cov()
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-20)
