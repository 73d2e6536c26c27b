"""Pins the CPU oracle to the committed mpmath known answers (tests/golden/kat_residual.json).

The reference ships no golden vectors, so these 50-digit evaluations of
TagReconstructionCostFunction.h:101-159 are what the restatement is checked against.
"""
import numpy as np


def test_residual_matches_mpmath(oracle, kats):
    for case in kats["obs"]:
        r = oracle.obs_eval(case["intr"], case["dist"], case["cam_qt"], case["tag_qt"], case["wh"],
                            case["px"], jac=False)
        # pixel coordinates are O(1e3): 1e-9 px is ~1e-12 relative on the projection
        np.testing.assert_allclose(r, case["residual"], rtol=0, atol=2e-9)


def test_tangent_jacobians_match_mpmath(oracle, kats):
    for case in kats["obs"]:
        _, Jc, Jt = oracle.obs_eval(case["intr"], case["dist"], case["cam_qt"], case["tag_qt"],
                                    case["wh"], case["px"])
        for J, ref in ((Jc, np.array(case["J_cam"])), (Jt, np.array(case["J_tag"]))):
            scale = np.abs(ref).max()
            np.testing.assert_allclose(J, ref, rtol=0, atol=1e-11 * scale)


def test_plus_matches_mpmath(oracle, kats):
    for case in kats["plus"]:
        np.testing.assert_allclose(oracle.pose_plus(case["qt"], case["delta"]), case["out"],
                                   rtol=0, atol=1e-15)


def test_huber_matches_closed_form(oracle, kats):
    for case in kats["huber"]:
        np.testing.assert_allclose(oracle.huber(case["a"], case["s"]), case["rho"], rtol=1e-15,
                                   atol=0)


def test_project_point_matches_camera_model_kats(oracle, kats):
    # CameraModel::projectPoint (CameraModel.cpp:6-26): pt.x() is overwritten before pt.y() is computed, so the
    # y tangential term sees the distorted x -- NOT the functor's formula (CostFunction.h:141-144).
    worst = 0.0
    for case in kats["project_point"]:
        uv = oracle.project_point(case["intr"], case["dist"], case["point_cam"])
        np.testing.assert_allclose(uv, case["uv"], rtol=0, atol=1e-9)
        worst = max(worst, abs(case["uv"][1] - case["uv_functor_formula"][1]))
    assert worst > 0.01   # the two formulas really differ at README distortion (up to 0.02 px)


def test_reprojection_statistics_match_camera_model_kats(oracle, kats):
    # computeReprojectionErrorPerCorner (src/TagReconstructor.cpp:430-455): Eigen rotations (no normalisation)
    # + CameraModel::projectPoint, one observation per KAT case
    for case in kats["obs"]:
        sc = oracle.Scene(case["intr"], case["dist"], [case["cam_qt"]], [case["tag_qt"]], [case["wh"]], -1,
                          [0], [0], [case["px"]])
        pc, pt, avg, corner = oracle.reprojection_stats(sc)
        ref = np.array(case["reprojection_error_camera_model"])
        np.testing.assert_allclose(corner[0], ref, rtol=0, atol=2e-9)
        mean = np.sqrt((ref.reshape(4, 2) ** 2).sum(axis=1)).sum() / 4
        np.testing.assert_allclose([pc[0], pt[0], avg], mean, rtol=1e-12)


def test_point_functor_matches_mpmath(oracle, kats):
    # OpenCVReprojectionError (TagReconstructionCostFunction.h:21-68): UnitQuaternionRotatePoint does not normalise
    # the camera quaternion -- cases 2 and 3 carry |q| = 1.05
    for case in kats["point_obs"]:
        r, Jc, Jp = oracle.point_eval(case["intr"], case["dist"], case["cam_qt"], case["point"], case["uv"])
        np.testing.assert_allclose(r, case["residual"], rtol=0, atol=2e-9)
        for J, ref in ((Jc, np.array(case["J_cam"])), (Jp, np.array(case["J_point"]))):
            np.testing.assert_allclose(J, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    # and it really differs from the normalising tag functor on a non-unit quaternion
    c = kats["point_obs"][2]
    tag = [1.0, 0.0, 0.0, 0.0] + list(c["point"])
    r_tag = oracle.obs_eval(c["intr"], c["dist"], c["cam_qt"], tag, [0.0, 0.0], list(c["uv"]) * 4, jac=False)[:2]
    assert np.abs(r_tag - np.array(c["residual"])).max() > 1.0
