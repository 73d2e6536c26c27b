"""CPU checks of the oracle's LM restatement (Ceres semantics).

PARITY UNPINNED against Ceres itself: no Ceres here and no reference fixtures.  What is checked:
the three exact linear solvers agree, a zero-noise scene converges to ground truth, and scipy's
independent minimiser cannot improve on the oracle's optimum.
"""
import numpy as np
import pytest

from visual_marker_mapping_amd.synthetic import make_scene


def _scene(O, s, init=True):
    return O.Scene(s.intr, s.dist, s.cam_init if init else s.cam_gt,
                   s.tag_init if init else s.tag_gt, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                   s.obs_px)


def test_linear_solvers_agree(oracle):
    s = make_scene(1)
    sols = []
    for solver in (oracle.DENSE_NORMAL, oracle.SCHUR_ELIM_TAGS, oracle.SCHUR_ELIM_CAMS):
        sc = _scene(oracle, s)
        summ, trace = oracle.solve(sc, oracle.default_options(robustify=1, linear_solver=solver))
        assert summ["termination_type"] == oracle.CONVERGENCE
        sols.append((summ, trace, sc.cam_qt.copy(), sc.tag_qt.copy()))
    for summ, trace, cq, tq in sols[1:]:
        assert summ["iterations"] == sols[0][0]["iterations"]
        np.testing.assert_allclose(cq, sols[0][2], rtol=0, atol=1e-10)
        np.testing.assert_allclose(tq, sols[0][3], rtol=0, atol=1e-10)
        for a, b in zip(trace, sols[0][1]):
            assert a["step_is_successful"] == b["step_is_successful"]
            np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-9)


def test_zero_noise_scene_recovers_ground_truth(oracle):
    s = make_scene(1, noise_px=0.0)
    sc = _scene(oracle, s)
    summ, _ = oracle.solve(sc, oracle.default_options(robustify=0, function_tolerance=1e-16,
                                                      parameter_tolerance=1e-14,
                                                      max_num_iterations=50))
    assert summ["final_cost"] < 1e-12

    def same_pose(a, b):
        sign = np.sign(np.sum(a[:, :4] * b[:, :4], axis=1))[:, None]
        np.testing.assert_allclose(a[:, :4] * sign, b[:, :4], rtol=0, atol=1e-9)
        np.testing.assert_allclose(a[:, 4:], b[:, 4:], rtol=0, atol=1e-9)

    same_pose(sc.cam_qt, s.cam_gt)
    same_pose(sc.tag_qt, s.tag_gt)


def test_origin_tag_stays_fixed_and_quaternions_stay_unit(oracle):
    s = make_scene(1)
    sc = _scene(oracle, s)
    oracle.solve(sc, oracle.default_options())
    np.testing.assert_array_equal(sc.tag_qt[0], s.tag_init[0])
    np.testing.assert_allclose(np.linalg.norm(sc.cam_qt[:, :4], axis=1), 1.0, atol=1e-12)


def test_scipy_cannot_improve_the_optimum(oracle):
    scipy_opt = pytest.importorskip("scipy.optimize")
    s = make_scene(1, n_cams=8, n_tags=5)
    sc = _scene(oracle, s)
    summ, _ = oracle.solve(sc, oracle.default_options(robustify=0, function_tolerance=1e-15,
                                                      parameter_tolerance=1e-13,
                                                      max_num_iterations=100))
    n_c, n_t = len(sc.cam_qt), len(sc.tag_qt)

    def residuals(d):
        d = d.reshape(-1, 6)
        out = []
        cams = [oracle.pose_plus(sc.cam_qt[c], d[c]) for c in range(n_c)]
        tags = [sc.tag_qt[0]] + [oracle.pose_plus(sc.tag_qt[t], d[n_c + t - 1])
                                 for t in range(1, n_t)]
        for i in range(len(sc.obs_cam)):
            out.append(oracle.obs_eval(sc.intr, sc.dist, cams[sc.obs_cam[i]], tags[sc.obs_tag[i]],
                                       sc.tag_wh[sc.obs_tag[i]], sc.obs_px[i], jac=False))
        return np.concatenate(out)

    x0 = np.zeros(6 * (n_c + n_t - 1))
    res = scipy_opt.least_squares(residuals, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    assert 0.5 * np.sum(res.fun ** 2) >= summ["final_cost"] * (1 - 1e-9)
    assert np.abs(res.x).max() < 1e-6


def test_robust_cost_is_per_corner_huber(oracle):
    # Huber acts on the squared norm of the 2-vector corner residual (TagReconstructor.cpp:721)
    s = make_scene(5, n_cams=6, n_tags=4)
    sc = _scene(oracle, s, init=False)
    # the functor's residuals (the statistics use CameraModel::projectPoint, a different formula with distortion)
    per_corner = np.array([oracle.obs_eval(sc.intr, sc.dist, sc.cam_qt[c], sc.tag_qt[t], sc.tag_wh[t], px, jac=False)
                           for c, t, px in zip(sc.obs_cam, sc.obs_tag, sc.obs_px)])
    sq = (per_corner.reshape(-1, 4, 2) ** 2).sum(axis=2)
    rho = np.where(sq > 1.0, 2.0 * np.sqrt(sq) - 1.0, sq)
    np.testing.assert_allclose(oracle.cost(sc, oracle.default_options(robustify=1)),
                               0.5 * rho.sum(), rtol=1e-12)
    np.testing.assert_allclose(oracle.cost(sc, oracle.default_options(robustify=0)),
                               0.5 * sq.sum(), rtol=1e-12)


def test_max_iterations_gives_no_convergence(oracle):
    s = make_scene(1)
    sc = _scene(oracle, s)
    summ, trace = oracle.solve(sc, oracle.default_options(max_num_iterations=2))
    assert summ["termination_type"] == oracle.NO_CONVERGENCE
    assert summ["iterations"] == 3 and trace[-1]["iteration"] == 2


def _point_scene(oracle, s, init=True):
    cam = s.cam_init if init else s.cam_gt
    tag = s.tag_init if init else s.tag_gt
    return oracle.point_scene(s.intr, s.dist, cam, tag, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)


def test_point_landmark_solvers_agree(oracle):
    """doBundleAdjustment_points (src/TagReconstructor.cpp:457-644): cameras x free 3-D points.  The dense normal
    equations and both block eliminations are three routes to the same LM step."""
    s = make_scene(1)
    res = []
    for solver in (oracle.DENSE_NORMAL, oracle.SCHUR_ELIM_TAGS, oracle.SCHUR_ELIM_CAMS):
        sc, _ = _point_scene(oracle, s)
        summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, linear_solver=solver))
        assert summ["termination_type"] == oracle.CONVERGENCE
        res.append((summ, trace, sc.cam_qt.copy(), oracle.scene_points(sc).copy()))
    for summ, trace, cam, pts in res[1:]:
        assert summ["iterations"] == res[0][0]["iterations"]
        np.testing.assert_allclose([t["cost"] for t in trace], [t["cost"] for t in res[0][1]], rtol=1e-9)
        np.testing.assert_allclose(cam, res[0][2], rtol=0, atol=1e-8)
        np.testing.assert_allclose(pts, res[0][3], rtol=0, atol=1e-8)
    # the origin tag's four corners are constant (:494-497), 8 residuals per tag observation, more freedom than the
    # tag-pose model (12 instead of 6 parameters per tag): a lower optimum on the same detections
    sc, pts0 = _point_scene(oracle, s)
    np.testing.assert_array_equal(res[0][3][s.fixed_tag], pts0[s.fixed_tag])
    sct = _scene(oracle, s)
    summ_t, _ = oracle.solve(sct, oracle.default_options(robustify=0))
    assert res[0][0]["final_cost"] < summ_t["final_cost"]
    assert res[0][0]["initial_cost"] == pytest.approx(summ_t["initial_cost"], rel=1e-9)   # same corners, same poses


def test_point_landmark_zero_noise_recovers_the_corners(oracle):
    s = make_scene(1, noise_px=0.0)
    sc, _ = _point_scene(oracle, s)
    summ, _ = oracle.solve(sc, oracle.default_options(robustify=0, function_tolerance=1e-16, parameter_tolerance=1e-14,
                                                      max_num_iterations=60))
    _, gt = _point_scene(oracle, s, init=False)
    assert summ["final_cost"] < 1e-10
    np.testing.assert_allclose(oracle.scene_points(sc), gt, rtol=0, atol=1e-8)


def test_rejected_steps_follow_ceres_policy(oracle):
    """A start 50 degrees / 0.8 m off the optimum: the trust region shrinks through consecutive rejected steps.
    TrustRegionMinimizer semantics on that path (Appendix A.4): a rejected step leaves x, the cost and the gradient
    norm untouched, divides the radius by a factor that doubles with every consecutive rejection (2, 4, 8, ...), and
    the next accepted step resets the factor; the three exact linear solvers take the same decisions."""
    s = make_scene(1, n_cams=20, n_tags=10, cam_rot_deg=50.0, cam_trans_m=0.8, tag_rot_deg=50.0, tag_trans_m=0.5)
    runs = []
    for solver in (oracle.DENSE_NORMAL, oracle.SCHUR_ELIM_TAGS, oracle.SCHUR_ELIM_CAMS):
        sc = _scene(oracle, s)
        summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, linear_solver=solver))
        assert summ["termination_type"] == oracle.CONVERGENCE and summ["num_unsuccessful_steps"] >= 3
        runs.append((summ, trace))
    summ, trace = runs[0]
    for other, tr in runs[1:]:
        assert other["iterations"] == summ["iterations"]
        assert [t["step_is_successful"] for t in tr] == [t["step_is_successful"] for t in trace]
        np.testing.assert_allclose([t["cost"] for t in tr], [t["cost"] for t in trace], rtol=1e-7)
    factor = 2.0
    for prev, cur in zip(trace, trace[1:]):
        if cur["step_is_successful"]:
            assert cur["cost"] < prev["cost"] or not prev["step_is_successful"]
            factor = 2.0
        else:
            # a row carries the radius AFTER its step's accept / reject update (IterationSummary::trust_region_radius)
            np.testing.assert_allclose(cur["trust_region_radius"], prev["trust_region_radius"] / factor, rtol=1e-12)
            factor *= 2.0
            assert cur["gradient_max_norm"] == prev["gradient_max_norm"]
