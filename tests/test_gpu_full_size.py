"""BASELINE.json configurations at their full sizes, through the C-ABI, against the CPU oracle's Schur path
(oracle/vmm_oracle.c, PARITY UNPINNED against Ceres) and against size-independent properties:

  configs[4]  500 x 200, README distortion, 0.5 px noise + 2 % outliers, Huber(1.0)    (trace + solution + statistics)
  configs[3]  2000 x 1000, f32 J^T J accumulation, f64 residuals / gradient / reduced system / LM decisions
  configs[1]  500 x 200 at 25 % visibility                                            (trace + solution)
"""
import numpy as np
import pytest

from test_gpu_solve import _assert_same_solution, _assert_same_trace

pytestmark = pytest.mark.gpu


def _expected_optimum(s, n_cam, n_tag):
    return 0.5 * s.noise_px ** 2 * (8 * s.n_obs - 6 * (n_cam + n_tag - 1))


def test_config5_full_size_matches_oracle(oracle):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(5)
    assert len(s.cam_init) == 500 and len(s.tag_init) == 200 and np.any(s.dist != 0.0) and s.robustify
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px)
    try:
        out = ba.solve(eng.default_options(robustify=1), trace_capacity=128)
        cam, tag = ba.get_state()
        pc, pt, avg, corner = ba.reprojection_stats()
    finally:
        ba.close()
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=1, num_threads=8))
    assert out["termination_type"] == eng.CONVERGENCE
    _assert_same_trace(out, summ, trace)
    _assert_same_solution(cam, tag, sc, s.tag_wh)
    # statistics at the oracle's converged state (CameraModel::projectPoint arithmetic, src/CameraModel.cpp:20-23)
    opc, opt_, oavg, ocorner = oracle.reprojection_stats(sc)
    np.testing.assert_allclose(pc, opc, rtol=1e-6)
    np.testing.assert_allclose(pt, opt_, rtol=1e-6)
    np.testing.assert_allclose(avg, oavg, rtol=1e-6)
    np.testing.assert_allclose(corner, ocorner, rtol=0, atol=1e-4)   # px; the converged poses agree far below 1e-6
    costs = [t["cost"] for t in out["trace"] if t["step_is_successful"]]
    assert all(b < a for a, b in zip(costs, costs[1:]))
    np.testing.assert_array_equal(tag[0], s.tag_init[0])


def test_sparse_visibility_full_size_matches_oracle(oracle):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(2, visibility=0.25)
    assert 20000 < s.n_obs < 30000
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px)
    try:
        out = ba.solve(eng.default_options(robustify=0), trace_capacity=128)
        cam, tag = ba.get_state()
    finally:
        ba.close()
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8))
    assert out["termination_type"] == eng.CONVERGENCE
    _assert_same_trace(out, summ, trace)
    _assert_same_solution(cam, tag, sc, s.tag_wh)
    assert abs(out["final_cost"] - _expected_optimum(s, 500, 200)) < 0.03 * _expected_optimum(s, 500, 200)


def test_f32_accumulate_mid_size_matches_oracle(oracle):
    """configs[3]'s precision at 400 x 250 (cameras eliminated, reduced order 1500 = 24 blocks: k_chol_dataflow with
    324 workgroups on 256 compute units, which must drain without a spin giving up): the f32 J^T J blocks perturb the Gauss-Newton model at 1e-7, so the trajectory follows the f64 oracle
    to ~1e-5 in cost and ends at the same optimum."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(4, n_cams=400, n_tags=250)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px, precision=eng.PRECISION_F32_ACCUM)
    try:
        out = ba.solve(eng.default_options(robustify=0), trace_capacity=128)
        cam, tag = ba.get_state()
    finally:
        ba.close()
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8))
    assert out["termination_type"] == summ["termination_type"] == eng.CONVERGENCE
    assert out["iterations"] == summ["iterations"]
    assert out["num_sync_timeouts"] == 0
    for a, b in zip(out["trace"], trace):
        assert a["step_is_successful"] == b["step_is_successful"]
        np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-5)
    np.testing.assert_allclose(out["final_cost"], summ["final_cost"], rtol=1e-7)
    scale = max(np.abs(sc.cam_qt).max(), np.abs(sc.tag_qt).max())
    np.testing.assert_allclose(cam, sc.cam_qt, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(tag, sc.tag_qt, rtol=0, atol=1e-5 * scale)
    exp = _expected_optimum(s, 400, 250)
    assert abs(out["final_cost"] - exp) < 0.02 * exp


_CONFIG4_FIRST = {}


def _config4_first_iteration(oracle, s):
    """The oracle's first LM iteration on configs[3] (Schur path, 8 threads): computed once per session."""
    if not _CONFIG4_FIRST:
        sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8, max_num_iterations=1))
        _CONFIG4_FIRST.update(sc=sc, summ=summ, trace=trace)
    return _CONFIG4_FIRST


def _assert_first_iteration(eng, ba, one, first):
    summ, trace, sc = first["summ"], first["trace"], first["sc"]
    assert one["iterations"] == summ["iterations"] == 2 and one["termination_type"] == eng.NO_CONVERGENCE
    np.testing.assert_allclose(one["trace"][0]["cost"], trace[0]["cost"], rtol=1e-12)
    np.testing.assert_allclose(one["trace"][0]["gradient_max_norm"], trace[0]["gradient_max_norm"], rtol=1e-9)
    assert one["trace"][1]["step_is_successful"] == trace[1]["step_is_successful"] == 1
    np.testing.assert_allclose(one["trace"][1]["cost"], trace[1]["cost"], rtol=1e-4)
    np.testing.assert_allclose(one["trace"][1]["model_cost_change"], trace[1]["model_cost_change"], rtol=1e-5)
    np.testing.assert_allclose(one["trace"][1]["step_norm"], trace[1]["step_norm"], rtol=1e-5)
    np.testing.assert_allclose(one["trace"][1]["trust_region_radius"], trace[1]["trust_region_radius"], rtol=1e-4)
    cam1, tag1 = ba.get_state()
    scale = max(np.abs(sc.cam_qt).max(), np.abs(sc.tag_qt).max())
    np.testing.assert_allclose(cam1, sc.cam_qt, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(tag1, sc.tag_qt, rtol=0, atol=1e-5 * scale)


@pytest.mark.parametrize("recover", [False, True])
def test_config4_first_iteration_block_sparse_elimination(oracle, monkeypatch, recover):
    """2000 x 1000 with the block-sparse elimination forced (VMM_BA_SCHUR=sparse: 1.0e9 pair terms, an 8 GB term list on the
    device, the reduced system of 94 block columns on the launch-per-column factorisation with the one-launch kernel on
    the last 34): the first LM iteration against the oracle.  Round 3 saw this configuration 0.29 % off once
    (gpurun_out/r3z/tests_all_nd.txt) and green minutes later without a recorded cause -- DESIGN.md section 4.11; this is
    its permanent test.  recover: the same with the one-launch kernels of the tail forced to give up, i.e. the pass
    redone from the rank-k update on (S is rebuilt by k_fill_lower + k_schur_pairs) -- the path that runs when the 629
    workgroups of the tail are not dispatched in order."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(4)
    first = _config4_first_iteration(oracle, s)
    monkeypatch.setenv("VMM_BA_SCHUR", "sparse")
    if recover:
        monkeypatch.setenv("VMM_BA_DEBUG_SPIN_LIMIT", "1")
        monkeypatch.setenv("VMM_BA_DEBUG_SPIN_KERNEL", "both")
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px,
                            precision=eng.PRECISION_F32_ACCUM)
    try:
        one = ba.solve(eng.default_options(robustify=0, max_num_iterations=1), trace_capacity=8)
        assert one["block_sparse"] == 1
        assert (one["num_sync_timeouts"] >= 1) == recover, one["num_sync_timeouts"]
        _assert_first_iteration(eng, ba, one, first)
    finally:
        ba.close()


def test_config4_full_size_f32_accumulate(oracle):
    """BASELINE.json configs[3]: 2000 images x 1000 tags (2 000 000 tag observations, reduced system of order
    6000), f32 J^T J accumulation.  Checked at full size: (i) the evaluation kernels against the oracle's
    per-observation functor on sampled observations and whole poses, (ii) the first LM iteration (iteration-0 and
    iteration-1 trace rows) against the oracle's Schur path, (iii) the converged solve against size-independent
    properties (convergence, monotone cost, expected optimum, unit quaternions, constant origin tag)."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(4)
    n_c, n_t = len(s.cam_init), len(s.tag_init)
    assert (n_c, n_t, s.n_obs) == (2000, 1000, 2000000)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px, precision=eng.PRECISION_F32_ACCUM)
    try:
        # (i) blocks at the initial state
        blk = ba.eval_blocks(robustify=False, want_W=True)
        rng = np.random.default_rng(4)
        sample = rng.choice(s.n_obs, 3000, replace=False)
        cost_s = 0.0
        for i in sample:
            c, t = s.obs_cam[i], s.obs_tag[i]
            r, Jc, Jt = oracle.obs_eval(s.intr, s.dist, s.cam_init[c], s.tag_init[t], s.tag_wh[t], s.obs_px[i])
            if t == s.fixed_tag:
                Jt = np.zeros_like(Jt)
            Wref = Jc.T @ Jt
            scale = max(np.abs(Jc).max() * np.abs(Jt).max(), 1e-300)
            assert np.abs(blk["W"][i] - Wref).max() <= 2e-6 * 8 * scale        # eight f32 products per entry
        for c in (0, 777, 1999):
            idx = np.nonzero(s.obs_cam == c)[0]
            V, g = np.zeros((6, 6)), np.zeros(6)
            for i in idx:
                t = s.obs_tag[i]
                r, Jc, _ = oracle.obs_eval(s.intr, s.dist, s.cam_init[c], s.tag_init[t], s.tag_wh[t], s.obs_px[i])
                V += Jc.T @ Jc
                g += Jc.T @ r
            assert np.abs(blk["V"][c] - V).max() <= 1e-5 * np.abs(V).max()     # f32 accumulation over 8000 rows
            np.testing.assert_allclose(blk["g_cam"][c], g, rtol=0, atol=1e-11 * np.abs(g).max())   # f64
        for t in (1, 500, 999):
            idx = np.nonzero(s.obs_tag == t)[0]
            U, g = np.zeros((6, 6)), np.zeros(6)
            for i in idx:
                c = s.obs_cam[i]
                r, _, Jt = oracle.obs_eval(s.intr, s.dist, s.cam_init[c], s.tag_init[t], s.tag_wh[t], s.obs_px[i])
                U += Jt.T @ Jt
                g += Jt.T @ r
            assert np.abs(blk["U"][t] - U).max() <= 1e-5 * np.abs(U).max()
            np.testing.assert_allclose(blk["g_tag"][t], g, rtol=0, atol=1e-11 * np.abs(g).max())
        assert np.all(blk["U"][s.fixed_tag] == 0.0)
        del blk
        sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                          s.obs_px)
        np.testing.assert_allclose(ba.cost(robustify=False), oracle.cost(sc, oracle.default_options(robustify=0)),
                                   rtol=1e-12)
        # (ii) one LM iteration against the oracle
        one = ba.solve(eng.default_options(robustify=0, max_num_iterations=1), trace_capacity=8)
        first = _config4_first_iteration(oracle, s)
        _assert_first_iteration(eng, ba, one, first)
        # (iii) the whole solve from the initial state
        ba.set_state(s.cam_init, s.tag_init)
        out = ba.solve(eng.default_options(robustify=0), trace_capacity=64)
        cam, tag = ba.get_state()
    finally:
        ba.close()
    assert out["termination_type"] == eng.CONVERGENCE
    costs = [t["cost"] for t in out["trace"] if t["step_is_successful"]]
    assert len(costs) >= 4 and all(b < a for a, b in zip(costs, costs[1:]))
    exp = _expected_optimum(s, n_c, n_t)
    assert abs(out["final_cost"] - exp) < 0.02 * exp
    np.testing.assert_array_equal(tag[s.fixed_tag], s.tag_init[s.fixed_tag])
    np.testing.assert_allclose(np.linalg.norm(cam[:, :4], axis=1), 1.0, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(tag[:, :4], axis=1), 1.0, atol=1e-12)
    # close to the ground truth the scene was generated from (0.3 px noise, 1000 tags per image)
    sign = np.sign(np.sum(cam[:, :4] * s.cam_gt[:, :4], axis=1))[:, None]
    assert np.abs(cam[:, :4] * sign - s.cam_gt[:, :4]).max() < 1e-3
    assert np.abs(cam[:, 4:] - s.cam_gt[:, 4:]).max() < 2e-2


@pytest.mark.parametrize("n_cams,n_tags", [(600, 320), (800, 400), (1000, 500)])
def test_one_launch_factorisation_beyond_residency_does_not_give_up(n_cams, n_tags):
    """Reduced systems of 30, 38 and 47 block columns: the one-launch factorisation runs with 495 to 1175 workgroups on 256
    compute units and drains only because the hardware dispatches workgroups in blockIdx order (a workgroup waits for
    lower-numbered ones only).  HIP does not promise that order; a wrong guess is a bounded spin giving up and the pass being
    redone -- correct, but 0.6 s per pass.  Until round 3 only the 24-block case asserted that no give-up happens
    (Weak #10).  Three LM iterations each, properties only (the trajectory at these sizes is covered by
    test_f32_accumulate_mid_size_matches_oracle and the kernel tests)."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(2, n_cams=n_cams, n_tags=n_tags)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    try:
        out = ba.solve(eng.default_options(robustify=0, max_num_iterations=3), trace_capacity=8)
    finally:
        ba.close()
    assert out["num_sync_timeouts"] == 0 and out["sync_timeout_kernels"] == 0
    assert out["num_lm_iterations"] == 3 and out["num_unsuccessful_steps"] == 0
    costs = [t["cost"] for t in out["trace"]]
    assert all(b < a for a, b in zip(costs, costs[1:]))
