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


def test_project_point_is_the_functor_tail(oracle, kats):
    # CameraModel::projectPoint (CameraModel.cpp:6-26) == CostFunction.h:125-152
    case = kats["obs"][1]
    uv = oracle.project_point(case["intr"], case["dist"], [0.3, -0.2, 2.5])
    x, y = 0.3 / 2.5, -0.2 / 2.5
    r2 = x * x + y * y
    k1, k2, p1, p2, k3 = case["dist"]
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    np.testing.assert_allclose(uv, [case["intr"][0] * xd + case["intr"][2],
                                    case["intr"][1] * yd + case["intr"][3]], rtol=1e-14)
