"""GPU parity on degenerate / ragged problems (the shapes startReconstruction produces on its way up):
one camera and one tag, poses without observations in the middle of the arrays, no constant block at all,
more tags than cameras, a zero-iteration budget, and empty observation lists.  Oracle = oracle/ (CPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6


def _solve_both(s, oracle, cam0, tag0, fixed, oc, ot, px, wh, **opt):
    from visual_marker_mapping_amd import engine as eng
    ba = eng.BundleAdjuster(s.intr, s.dist, cam0, tag0, wh, fixed, oc, ot, px)
    out = ba.solve(eng.default_options(**opt))
    cam, tag = ba.get_state()
    pc, pt, avg, _ = ba.reprojection_stats()
    ba.close()
    sc = oracle.Scene(s.intr, s.dist, cam0, tag0, wh, fixed, oc, ot, px)
    summ, _ = oracle.solve(sc, oracle.default_options(linear_solver=oracle.DENSE_NORMAL, **opt))
    opc, opt_, oavg, _ = oracle.reprojection_stats(sc)
    assert out["termination_type"] == summ["termination_type"] and out["iterations"] == summ["iterations"]
    scale = max(np.abs(sc.cam_qt).max(), np.abs(sc.tag_qt).max())
    np.testing.assert_allclose(cam, sc.cam_qt, rtol=0, atol=REL * scale)
    np.testing.assert_allclose(tag, sc.tag_qt, rtol=0, atol=REL * scale)
    np.testing.assert_allclose(pc, opc, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(pt, opt_, rtol=1e-6, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(avg, oavg, rtol=1e-6)
    return out, cam, tag, pc, pt


def test_one_camera_one_fixed_tag(oracle):
    """The first image of startReconstruction: only the camera moves; the tag block is constant."""
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=1, n_tags=1)
    out, cam, tag, _, _ = _solve_both(s, oracle, s.cam_init, s.tag_init, 0, s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh,
                                      robustify=1)
    assert np.array_equal(tag, s.tag_init)            # constant block untouched, bit for bit
    assert out["final_cost"] < out["initial_cost"]


def test_poses_without_observations_stay_put(oracle):
    """Camera 2 and tag 3 lose all their observations: they are not part of the reduced program
    (src/TagReconstructor.cpp:689-690 never adds such a camera), keep their values, and the statistics report
    -1.0 for the camera (:379-383) and NaN (no map entry) for the tag."""
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=7, n_tags=6)
    keep = (s.obs_cam != 2) & (s.obs_tag != 3)
    out, cam, tag, pc, pt = _solve_both(s, oracle, s.cam_init, s.tag_init, 0, s.obs_cam[keep], s.obs_tag[keep],
                                        s.obs_px[keep], s.tag_wh, robustify=0)
    assert np.array_equal(cam[2], s.cam_init[2]) and np.array_equal(tag[3], s.tag_init[3])
    assert pc[2] == -1.0 and np.isnan(pt[3])
    assert not np.array_equal(cam[1], s.cam_init[1])


def test_no_constant_block(oracle):
    """fixed_tag = -1: the gauge is free, the normal matrix is singular without damping; LM's diagonal
    makes every step well defined and both implementations walk the same path."""
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=6, n_tags=5)
    _solve_both(s, oracle, s.cam_init, s.tag_init, -1, s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, robustify=0,
                max_num_iterations=12)


def test_more_tags_than_cameras_eliminates_tags(oracle):
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=4, n_tags=23, visibility=0.8)
    _solve_both(s, oracle, s.cam_init, s.tag_init, 0, s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, robustify=1)


def test_zero_iterations_and_empty_problem(oracle):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=3, n_tags=3)
    out, cam, tag, _, _ = _solve_both(s, oracle, s.cam_init, s.tag_init, 0, s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh,
                                      robustify=0, max_num_iterations=0)
    assert out["termination_type"] == eng.NO_CONVERGENCE and np.array_equal(cam, s.cam_init)
    # no observations at all: nothing to optimise, poses untouched, every camera reports -1
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, [], [], np.zeros((0, 8)))
    try:
        out = ba.solve(eng.default_options())
        cam, tag = ba.get_state()
        pc, pt, avg, corner = ba.reprojection_stats()
        assert np.all(ba.tag_translation_covariance() == 0.0)
    finally:
        ba.close()
    assert out["termination_type"] == eng.CONVERGENCE and np.array_equal(cam, s.cam_init) and np.array_equal(tag, s.tag_init)
    assert np.all(pc == -1.0) and np.all(np.isnan(pt)) and corner.shape == (0, 8)


def test_observation_mask_equals_a_rebuilt_subproblem(oracle):
    """vmm_ba_set_observation_mask: one handle for the whole detection set, a mask per step of the incremental
    driver.  Switching off every observation of cameras 5.. and of tag 3 must give the solve, the statistics and
    the covariance of the handle built from the remaining observations only (poses that lose all observations
    drop out like poses without observations)."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(5, n_cams=9, n_tags=7, visibility=0.8)
    keep = (s.obs_cam < 5) & (s.obs_tag != 3)
    full = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
    sub = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam[keep], s.obs_tag[keep],
                             s.obs_px[keep])
    try:
        full.set_observation_mask(keep)
        assert full.cost(robustify=True) == pytest.approx(sub.cost(robustify=True), rel=1e-12)
        o = eng.default_options(robustify=1)
        a, b = full.solve(o, trace_capacity=64), sub.solve(o, trace_capacity=64)
        assert a["iterations"] == b["iterations"] and a["termination_type"] == b["termination_type"]
        for x, y in zip(a["trace"], b["trace"]):
            assert x["step_is_successful"] == y["step_is_successful"]
            np.testing.assert_allclose(x["cost"], y["cost"], rtol=1e-9)
        (ca, ta), (cb, tb) = full.get_state(), sub.get_state()
        np.testing.assert_allclose(ca, cb, rtol=0, atol=1e-9 * np.abs(cb).max())
        np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-9 * np.abs(tb).max())
        assert np.array_equal(ca[5:], s.cam_init[5:]) and np.array_equal(ta[3], s.tag_init[3])   # switched off: untouched
        pa, pb = full.reprojection_stats(), sub.reprojection_stats()
        np.testing.assert_allclose(pa[0], pb[0], rtol=1e-9)
        np.testing.assert_allclose(pa[1], pb[1], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(pa[2], pb[2], rtol=1e-9)
        assert np.all(pa[0][5:] == -1.0) and np.isnan(pa[1][3])
        # per-corner errors in pixels (coordinates of a few thousand): the two solves agree to 1e-9 of the pose scale
        np.testing.assert_allclose(pa[3][keep], pb[3], rtol=1e-9, atol=1e-9)
        assert np.all(pa[3][~keep] == 0.0)
        np.testing.assert_allclose(full.tag_translation_covariance(True), sub.tag_translation_covariance(True),
                                   rtol=1e-6, atol=1e-18)
        # back to all observations: the same as a handle that never had a mask
        full.set_observation_mask(None)
        full.set_state(s.cam_init, s.tag_init)
        ref = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
        try:
            a, b = full.solve(o), ref.solve(o)
            assert a["iterations"] == b["iterations"] and a["final_cost"] == b["final_cost"]
        finally:
            ref.close()
        with pytest.raises(ValueError):
            full.set_observation_mask(keep[:-1])
    finally:
        full.close()
        sub.close()


def test_masked_observation_on_the_camera_plane_does_not_poison_the_solve(oracle):
    """A switched-off observation whose parked poses put the tag centre exactly on the camera plane (Z_c == 0:
    identity camera at the origin, tag at z = 0) would give 1/Z = inf; it must be selected out, not multiplied
    by zero (0 * inf = NaN in the cost, g, H, W).  The solve must equal the one without that observation."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, n_cams=6, n_tags=5)
    cam0 = np.vstack([s.cam_init, [[1.0, 0, 0, 0, 0, 0, 0]]])          # parked camera: identity at the origin
    tag0 = np.vstack([s.tag_init, [[1.0, 0, 0, 0, 0.3, -0.2, 0.0]]])   # parked tag on that camera's plane z = 0
    wh = np.vstack([s.tag_wh, s.tag_wh[:1]])
    oc = np.concatenate([s.obs_cam, [6, 6, 0]]).astype(np.int32)
    ot = np.concatenate([s.obs_tag, [5, 0, 5]]).astype(np.int32)
    px = np.vstack([s.obs_px, np.full((3, 8), 100.0)])
    mask = np.ones(len(oc), np.uint8)
    mask[-3:] = 0
    ba = eng.BundleAdjuster(s.intr, s.dist, cam0, tag0, wh, 0, oc, ot, px)
    ref = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
    try:
        ba.set_observation_mask(mask)
        assert np.isfinite(ba.cost(robustify=True))
        blocks = ba.eval_blocks(robustify=True)
        assert all(np.all(np.isfinite(blocks[k])) for k in ("V", "U", "W", "g_cam", "g_tag"))
        assert np.all(blocks["W"][-3:] == 0.0)
        o = eng.default_options(robustify=1)
        a, b = ba.solve(o, trace_capacity=64), ref.solve(o, trace_capacity=64)
        assert a["termination_type"] == b["termination_type"] == eng.CONVERGENCE
        assert a["iterations"] == b["iterations"] and a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-12)
        (ca, ta), (cb, tb) = ba.get_state(), ref.get_state()
        np.testing.assert_allclose(ca[:6], cb, rtol=0, atol=1e-10)
        np.testing.assert_allclose(ta[:5], tb, rtol=0, atol=1e-10)
        assert np.array_equal(ca[6], cam0[6]) and np.array_equal(ta[5], tag0[5])
    finally:
        ba.close()
        ref.close()
