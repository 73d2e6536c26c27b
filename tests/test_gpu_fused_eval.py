"""One evaluation per observation (k_eval_fused) against the two-pass kernel (k_eval_both) and the oracle.

Ceres evaluates TagReconstructionCostFunction::operator() (include/visual_marker_mapping/
TagReconstructionCostFunction.h:101-159) once per residual block; the two-pass kernel evaluates every observation
twice (once per pose family's order).  The fused kernel must produce the same blocks and the same LM trajectory,
bit-repeatably, for both eliminations, both precisions, point landmarks, masked observations and scenes where some
(camera, tag) pairs are not observed (idle lanes).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blocks_and_solve(s, monkeypatch, mode, elim, robust, precision=None, landmarks=None, mask=None):
    from visual_marker_mapping_amd import engine as eng
    monkeypatch.setenv("VMM_BA_EVAL", mode)
    kw = {}
    if precision is not None:
        kw["precision"] = precision
    if landmarks is not None:
        kw["landmarks"] = landmarks
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px, elimination=elim, **kw)
    try:
        if mask is not None:
            ba.set_observation_mask(mask)
        b = ba.eval_blocks(robustify=bool(robust))
        b2 = ba.eval_blocks(robustify=bool(robust))
        out = ba.solve(eng.default_options(robustify=robust), trace_capacity=128)
        cam, tag = ba.get_state()
    finally:
        ba.close()
    for k in ("cost", "V", "U", "W", "g_cam", "g_tag"):
        if b[k] is not None:
            assert np.array_equal(b[k], b2[k]), k          # bit-repeatable
    return b, out, cam, tag


CASES = {
    "dense": (dict(config=1), {}),
    "robust_distorted": (dict(config=5, n_cams=70, n_tags=40), {}),
    "holes": (dict(config=1, n_cams=50, n_tags=70, visibility=0.7), {}),       # idle lanes, two chunks of kept poses
    "f32": (dict(config=1, n_cams=40, n_tags=16), dict(precision="f32")),
    "points": (dict(config=1, n_cams=12, n_tags=9), dict(landmarks="points")),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("elim", ["cams", "tags"])
def test_fused_evaluation_equals_two_pass(monkeypatch, name, elim):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    skw, extra = CASES[name]
    skw = dict(skw)
    s = make_scene(skw.pop("config"), **skw)
    robust = 1 if s.robustify else 0
    mode = eng.ELIM_CAMERAS if elim == "cams" else eng.ELIM_TAGS
    kw = {}
    if extra.get("precision") == "f32":
        kw["precision"] = eng.PRECISION_F32_ACCUM
    if extra.get("landmarks") == "points":
        kw["landmarks"] = eng.LANDMARK_POINTS
        robust = 0
    mask = None
    if name == "holes":
        mask = np.ones(s.n_obs, np.uint8)
        mask[::7] = 0
    bt, ot, ct, tt = _blocks_and_solve(s, monkeypatch, "twopass", mode, robust, mask=mask, **kw)
    bf, of, cf, tf = _blocks_and_solve(s, monkeypatch, "fused", mode, robust, mask=mask, **kw)
    rel = 1e-5 if "precision" in kw else 1e-11    # f32 J^T J: the sums associate differently
    np.testing.assert_allclose(bf["cost"], bt["cost"], rtol=1e-13)
    for k in ("V", "U", "W", "g_cam", "g_tag"):
        if bt[k] is None:
            continue
        scale = np.abs(bt[k]).max()
        np.testing.assert_allclose(bf[k], bt[k], rtol=0, atol=rel * scale, err_msg=k)
    assert of["termination_type"] == ot["termination_type"] and of["iterations"] == ot["iterations"]
    for x, y in zip(of["trace"], ot["trace"]):
        assert x["step_is_successful"] == y["step_is_successful"]
        np.testing.assert_allclose(x["cost"], y["cost"], rtol=1e-5 if "precision" in kw else 1e-10)
    tol = 1e-5 if "precision" in kw else 1e-9
    np.testing.assert_allclose(cf, ct, rtol=0, atol=tol * np.abs(ct).max())
    np.testing.assert_allclose(tf, tt, rtol=0, atol=tol * np.abs(tt).max())


def test_fused_blocks_match_the_oracle(oracle, monkeypatch):
    """Blocks of the fused kernel against the oracle's functor at 1e-10 (full visibility: the default path)."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    monkeypatch.delenv("VMM_BA_EVAL", raising=False)
    s = make_scene(5, n_cams=90, n_tags=70)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    try:
        b = ba.eval_blocks(robustify=True)
    finally:
        ba.close()
    from test_gpu_kernels import _blocks_from_oracle
    ob = _blocks_from_oracle(oracle, s, s.cam_init, s.tag_init, True)
    np.testing.assert_allclose(b["cost"], ob["cost"], rtol=1e-12)
    for k in ("V", "U", "W", "g_cam", "g_tag"):
        scale = np.abs(ob[k]).max()
        np.testing.assert_allclose(b[k], ob[k], rtol=0, atol=1e-10 * scale, err_msg=k)
