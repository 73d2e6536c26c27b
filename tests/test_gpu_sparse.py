"""Block-sparse elimination (compressed Z, k_schur_pairs) against the dense path and the oracle.

The reference solves with Ceres' sparse normal Cholesky under its ordering (src/TagReconstructor.cpp:725-738) and real
projects see a handful of tags per image (README.md:155-216).  The dense path stores Z with its zero blocks and
multiplies them; the block-sparse path keeps only the 6x6 blocks of co-observed (camera, tag) pairs and forms
S(f, f') -= Z_ef^T Z_ef' over the pairs that share an eliminated pose.  Both are exact: same LM trajectory.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(s, robust, schur, monkeypatch, elim=None, want_cov=False):
    from visual_marker_mapping_amd import engine as eng
    if schur is None:
        monkeypatch.delenv("VMM_BA_SCHUR", raising=False)
    else:
        monkeypatch.setenv("VMM_BA_SCHUR", schur)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag,
                            s.obs_px, elimination=eng.ELIM_AUTO if elim is None else elim)
    try:
        out = ba.solve(eng.default_options(robustify=robust), trace_capacity=256)
        cam, tag = ba.get_state()
        cov = ba.tag_translation_covariance(robustify=bool(robust)) if want_cov else None
        out2 = None
        if want_cov:   # the covariance switched a sparse handle to the dense path and back: it still solves the same way
            ba.set_state(s.cam_init, s.tag_init)
            out2 = ba.solve(eng.default_options(robustify=robust))
    finally:
        ba.close()
    return out, cam, tag, cov, out2


def _assert_same_run(a, b, ca, ta, cb, tb, rtol=1e-10):
    assert a["termination_type"] == b["termination_type"] and a["iterations"] == b["iterations"]
    for x, y in zip(a["trace"], b["trace"]):
        assert x["step_is_successful"] == y["step_is_successful"]
        np.testing.assert_allclose(x["cost"], y["cost"], rtol=rtol)
        np.testing.assert_allclose(x["trust_region_radius"], y["trust_region_radius"], rtol=1e-6)
    np.testing.assert_allclose(ca, cb, rtol=0, atol=1e-9 * np.abs(cb).max())
    np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-9 * np.abs(tb).max())


SCENES = {
    "quarter": dict(config=5, n_cams=40, n_tags=30, visibility=0.25),
    "close_up": dict(config=1, n_cams=60, n_tags=40, neighbors_min=4, neighbors_max=7),
    "dense": dict(config=1),
    "far": dict(config=1, n_cams=30, n_tags=20, visibility=0.4, cam_rot_deg=50.0, cam_trans_m=0.8, tag_rot_deg=50.0,
                tag_trans_m=0.5),
    # kept family of 250 poses: two column groups in k_schur_rows
    "two_groups": dict(config=1, n_cams=300, n_tags=250, visibility=0.06),
}


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("elim", ["cams", "tags"])
def test_sparse_path_equals_dense_path(monkeypatch, name, elim):
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    kw = dict(SCENES[name])
    s = make_scene(kw.pop("config"), **kw)
    robust = 1 if s.robustify else 0
    mode = eng.ELIM_CAMERAS if elim == "cams" else eng.ELIM_TAGS
    d, cd, td, covd, _ = _solve(s, robust, "dense", monkeypatch, mode, want_cov=name == "quarter")
    sp, cs, ts, covs, again = _solve(s, robust, "sparse", monkeypatch, mode, want_cov=name == "quarter")
    assert d["block_sparse"] == 0 and sp["block_sparse"] == 1
    assert d["termination_type"] == eng.CONVERGENCE
    if name == "far":
        assert d["num_unsuccessful_steps"] >= 1
    _assert_same_run(sp, d, cs, ts, cd, td)
    if covd is not None:
        np.testing.assert_allclose(covs, covd, rtol=1e-7, atol=1e-20)
        assert again["iterations"] == sp["iterations"] and again["final_cost"] == sp["final_cost"]
    # fixed summation order: a second run is the same to the bit
    sp2, cs2, ts2, _, _ = _solve(s, robust, "sparse", monkeypatch, mode)
    assert sp2["final_cost"] == sp["final_cost"] and np.array_equal(cs, cs2) and np.array_equal(ts, ts2)


def test_path_choice_follows_the_block_structure(monkeypatch):
    """Full visibility keeps the dense MFMA update, a quarter of the pairs or a handful of tags per image goes sparse."""
    from visual_marker_mapping_amd.synthetic import make_scene
    for kw, want in ((dict(), 0), (dict(visibility=0.25), 1), (dict(neighbors_min=6, neighbors_max=10), 1)):
        s = make_scene(2, **kw)
        out, _, _, _, _ = _solve(s, 0, None, monkeypatch)
        assert out["block_sparse"] == want, (kw, out["block_sparse"])


def test_close_up_full_size_matches_oracle(oracle, monkeypatch):
    """500 images x 200 tags, every image sees the 6..10 tags nearest to the wall point it stands in front of (4034
    tag observations, a block-sparse reduced system): trace and solution against the oracle."""
    from test_gpu_solve import _assert_same_solution, _assert_same_trace
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(2, neighbors_min=6, neighbors_max=10)
    deg = np.bincount(s.obs_cam, minlength=500)
    assert deg.min() >= 6 and deg.max() <= 12 and np.bincount(s.obs_tag, minlength=200).min() >= 2
    out, cam, tag, _, _ = _solve(s, 0, None, monkeypatch)
    assert out["block_sparse"] == 1 and out["termination_type"] == eng.CONVERGENCE
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8))
    _assert_same_trace(out, summ, trace)
    _assert_same_solution(cam, tag, sc, s.tag_wh)
    np.testing.assert_array_equal(tag[0], s.tag_init[0])


def test_tree_ordering_of_the_kept_family(oracle, monkeypatch):
    """The close-up scene at full size: the handle orders its 200 tags by a nested dissection of the co-observation
    graph (summary.tree_ordering = number of tree nodes), every node on a 64-row boundary of the reduced system; the
    one-launch factorisation and the back-substitution follow the block structure of the factor.  Same LM trajectory as
    the natural order (VMM_BA_ORDER=natural) and as the oracle; the covariance (dense, natural order inside) and a
    forced give-up of both one-launch kernels (redone on the launch-per-column path) go through the same handle."""
    from test_gpu_solve import _assert_same_solution, _assert_same_trace
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(2, neighbors_min=6, neighbors_max=10)
    monkeypatch.delenv("VMM_BA_ORDER", raising=False)
    tree, cam_t, tag_t, cov_t, again_t = _solve(s, 0, None, monkeypatch, want_cov=True)
    assert tree["block_sparse"] == 1 and tree["tree_ordering"] >= 3 and tree["num_sync_timeouts"] == 0
    assert again_t["iterations"] == tree["iterations"] and again_t["final_cost"] == tree["final_cost"]
    monkeypatch.setenv("VMM_BA_ORDER", "natural")
    nat, cam_n, tag_n, cov_n, _ = _solve(s, 0, None, monkeypatch, want_cov=True)
    assert nat["block_sparse"] == 1 and nat["tree_ordering"] == 0
    _assert_same_run(tree, nat, cam_t, tag_t, cam_n, tag_n, rtol=1e-9)
    for t in range(1, len(tag_t)):
        np.testing.assert_allclose(cov_t[t], cov_n[t], rtol=0, atol=1e-7 * np.abs(cov_n[t]).max())
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8))
    _assert_same_trace(tree, summ, trace)
    _assert_same_solution(cam_t, tag_t, sc, s.tag_wh)
    # every pass's factorisation gives up and is redone without inter-workgroup waits: same trajectory
    monkeypatch.delenv("VMM_BA_ORDER", raising=False)
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_LIMIT", "1")
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_KERNEL", "both")
    monkeypatch.setenv("VMM_BA_DEBUG_SPIN_ONCE", "0")
    redo, cam_r, tag_r, _, _ = _solve(s, 0, None, monkeypatch)
    assert redo["tree_ordering"] == tree["tree_ordering"] and redo["num_sync_timeouts"] >= 1
    _assert_same_run(redo, tree, cam_r, tag_r, cam_t, tag_t, rtol=1e-9)
    # a robust solve with masked observations on the tree-ordered handle against the natural order
    monkeypatch.delenv("VMM_BA_DEBUG_SPIN_LIMIT")
    monkeypatch.delenv("VMM_BA_DEBUG_SPIN_KERNEL")
    monkeypatch.delenv("VMM_BA_DEBUG_SPIN_ONCE")
    mask = np.ones(s.n_obs, np.uint8)
    mask[::7] = 0
    res = []
    for order in (None, "natural"):
        if order:
            monkeypatch.setenv("VMM_BA_ORDER", order)
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        try:
            ba.set_observation_mask(mask)
            out = ba.solve(eng.default_options(robustify=1), trace_capacity=256)
            res.append((out,) + ba.get_state())
        finally:
            ba.close()
    _assert_same_run(res[0][0], res[1][0], res[0][1], res[0][2], res[1][1], res[1][2], rtol=1e-9)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tree_ordering_random_close_up_scenes(monkeypatch, seed):
    """Walls and corridors (tags in 1, 2 or 4 rows) of 70 to 380 tags seen 3..14 at a time (some tags unseen, the graph in several pieces now and then), both
    eliminations: forced tree ordering against the natural order -- the same LM trajectory."""
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    rng = np.random.default_rng(100 + seed)
    n_tags = int(rng.integers(70, 380))
    n_cams = int(rng.integers(n_tags // 2, 2 * n_tags))
    lo = int(rng.integers(3, 8))
    s = make_scene(5 if seed % 2 else 1, n_cams=n_cams, n_tags=n_tags, neighbors_min=lo, neighbors_max=lo + int(rng.integers(0, 7)),
                   wall_rows=(0, 1, 2, 0, 4, 0)[seed])   # walls and corridors
    robust = seed % 2
    elim = eng.ELIM_CAMERAS if seed % 3 else eng.ELIM_TAGS
    runs = []
    for order in ("nd", "natural"):
        monkeypatch.setenv("VMM_BA_ORDER", order)
        monkeypatch.setenv("VMM_BA_SCHUR", "sparse")
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px,
                                elimination=elim)
        try:
            out = ba.solve(eng.default_options(robustify=robust, max_num_iterations=12), trace_capacity=64)
            runs.append((out,) + ba.get_state())
        finally:
            ba.close()
    assert runs[1][0]["tree_ordering"] == 0 and runs[0][0]["num_sync_timeouts"] == 0
    _assert_same_run(runs[0][0], runs[1][0], runs[0][1], runs[0][2], runs[1][1], runs[1][2], rtol=1e-8)
    print("seed %d: %d cams x %d tags, elimination %s, tree nodes %d" % (seed, n_cams, n_tags, "cams" if elim == eng.ELIM_CAMERAS else "tags",
                                                                         runs[0][0]["tree_ordering"]))


def test_tree_ordering_at_2000_x_1000(oracle, monkeypatch):
    """The realistic LARGE scene: 2000 images x 1000 tags on a wall, every image sees its 6..10 nearest tags (about 16 000 tag
    observations).  Its reduced system of 1000 kept poses is as sparse as the small close-up scene's; the reference factors
    it with a fill-reducing ordering whatever the size (Ceres' sparse normal Cholesky, src/TagReconstructor.cpp:725-738).
    Round 3 took the tree ordering only up to 48 natural block columns: this scene got the dense 94-column factorisation
    (3.0 ms).  Now only the non-zero blocks of the factor have a workgroup (109 block columns, ~1300 workgroups, masks of 256
    bits per block row) and the one-launch kernel takes it.  Checked: the first two LM iterations against the oracle, the whole
    trajectory against the natural order (the launch-per-column factorisation of the same handle type), no give-up."""
    from test_gpu_solve import _assert_same_solution
    from visual_marker_mapping_amd import engine as eng
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(4, neighbors_min=6, neighbors_max=10)
    assert (len(s.cam_init), len(s.tag_init)) == (2000, 1000) and 10000 < s.n_obs < 25000
    monkeypatch.delenv("VMM_BA_ORDER", raising=False)
    tree, cam_t, tag_t, _, _ = _solve(s, 0, None, monkeypatch)
    assert tree["block_sparse"] == 1 and tree["tree_ordering"] >= 8 and tree["num_sync_timeouts"] == 0
    assert tree["termination_type"] == eng.CONVERGENCE
    # the default at this size is the six-wave form of the kernel (two helper workers while earlier panels are applied,
    # VMM_BA_DF_HELP): a tile sees the same MFMAs in the same order whoever issues them -- the four-wave form gives the same bits
    monkeypatch.setenv("VMM_BA_DF_HELP", "0")
    four, cam_4, tag_4, _, _ = _solve(s, 0, None, monkeypatch)
    monkeypatch.delenv("VMM_BA_DF_HELP", raising=False)
    assert four["iterations"] == tree["iterations"] and four["final_cost"] == tree["final_cost"]
    np.testing.assert_array_equal(cam_4, cam_t)
    np.testing.assert_array_equal(tag_4, tag_t)
    monkeypatch.setenv("VMM_BA_ORDER", "natural")
    nat, cam_n, tag_n, _, _ = _solve(s, 0, None, monkeypatch)
    assert nat["tree_ordering"] == 0 and nat["block_sparse"] == 1
    _assert_same_run(tree, nat, cam_t, tag_t, cam_n, tag_n, rtol=1e-9)
    # the first two LM iterations against the oracle (its Schur path on the dense 6000 x 6000 reduced system)
    monkeypatch.delenv("VMM_BA_ORDER", raising=False)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    try:
        two = ba.solve(eng.default_options(robustify=0, max_num_iterations=2), trace_capacity=8)
        cam2, tag2 = ba.get_state()
    finally:
        ba.close()
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    summ, trace = oracle.solve(sc, oracle.default_options(robustify=0, num_threads=8, max_num_iterations=2))
    assert two["iterations"] == summ["iterations"] == 3
    for k in range(3):
        assert two["trace"][k]["step_is_successful"] == trace[k]["step_is_successful"]
        np.testing.assert_allclose(two["trace"][k]["cost"], trace[k]["cost"], rtol=1e-8)
        np.testing.assert_allclose(two["trace"][k]["trust_region_radius"], trace[k]["trust_region_radius"], rtol=1e-6)
    _assert_same_solution(cam2, tag2, sc, s.tag_wh)
