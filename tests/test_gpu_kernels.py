"""GPU unit tests of the individual HIP kernels, called through the C-ABI (include/vmm_ba.h).

Run on the MI355X box with `pytest -m gpu`.  The checker is the CPU oracle (oracle/) or numpy; the
HIP path itself never touches either.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from visual_marker_mapping_amd import engine
    return engine


def _blocks_from_oracle(O, s, cam, tag, robustify):
    """V, U, W, g, cost from the oracle's per-observation residuals/Jacobians (numpy accumulation)."""
    n_c, n_t = len(cam), len(tag)
    V, U = np.zeros((n_c, 6, 6)), np.zeros((n_t, 6, 6))
    W = np.zeros((len(s.obs_cam), 6, 6))
    gc, gt = np.zeros((n_c, 6)), np.zeros((n_t, 6))
    cost = 0.0
    for i, (c, t) in enumerate(zip(s.obs_cam, s.obs_tag)):
        r, Jc, Jt = O.obs_eval(s.intr, s.dist, cam[c], tag[t], s.tag_wh[t], s.obs_px[i])
        if t == s.fixed_tag:
            Jt = np.zeros_like(Jt)
        for k in range(4):
            sq = r[2 * k] ** 2 + r[2 * k + 1] ** 2
            rho = O.huber(1.0, sq) if robustify else np.array([sq, 1.0, 0.0])
            cost += 0.5 * rho[0]
            w = np.sqrt(rho[1])
            Jc[2 * k:2 * k + 2] *= w
            Jt[2 * k:2 * k + 2] *= w
            r[2 * k:2 * k + 2] *= w
        V[c] += Jc.T @ Jc
        U[t] += Jt.T @ Jt
        W[i] = Jc.T @ Jt
        gc[c] += Jc.T @ r
        gt[t] += Jt.T @ r
    return dict(V=V, U=U, W=W, g_cam=gc, g_tag=gt, cost=cost)


def test_mfma_f64_syrk_matches_numpy(eng):
    rng = np.random.default_rng(1)
    for k, n in [(16, 64), (100, 130), (600, 200), (37, 65)]:
        Z = rng.standard_normal((k, n))
        # asymmetric, integer-valued data also catches a transposed C/D lane map exactly
        Zi = np.round(Z * 3.0)
        for M in (Z, Zi):
            Cg = eng.dense_syrk(M)
            ref = M.T @ M
            np.testing.assert_allclose(Cg, ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("wide", ["1", "0"])
def test_syrk_few_tiles_both_kernels(eng, monkeypatch, wide):
    """Few tiles (at most 64): k_syrk_wide (one 8-wave workgroup per CU, balanced diagonal tiles, in-workgroup sum of the
    two K halves) and, with VMM_BA_SYRK_WIDE=0, the two-workgroups-per-CU kernel -- the headline shape 3000 x 1217, K ranges
    of odd length, a single 16-row stage, a lone diagonal tile."""
    monkeypatch.setenv("VMM_BA_SYRK_WIDE", wide)
    rng = np.random.default_rng(77)
    for k, n in [(3000, 1217), (330, 520), (16, 128), (48, 1000), (1000, 129)]:
        Zi = np.round(rng.standard_normal((k, n)) * 3.0)   # integer-valued: exact whatever the summation order
        np.testing.assert_array_equal(eng.dense_syrk(Zi), Zi.T @ Zi)
        Z = rng.standard_normal((k, n))
        C = eng.dense_syrk(Z)
        ref = Z.T @ Z
        np.testing.assert_allclose(C, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
        np.testing.assert_array_equal(C, eng.dense_syrk(Z))


def test_blocked_cholesky_solve_matches_numpy(eng):
    rng = np.random.default_rng(2)
    for n in (5, 64, 65, 200, 333, 1200):
        B = rng.standard_normal((n, n))
        A = B @ B.T + n * np.eye(n)
        b = rng.standard_normal(n)
        x, info = eng.dense_spd_solve(A, b)
        assert info == 0
        ref = np.linalg.solve(A, b)
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-10 * np.abs(ref).max())


def test_cholesky_solve_is_repeatable(eng):
    """The factorisation runs 19+ dependent launches and a flag-chained back-substitution: repeat it to
    catch any ordering hazard (every run must give the same bits)."""
    rng = np.random.default_rng(5)
    n = 700
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x0, info0 = eng.dense_spd_solve(A, b)
    assert info0 == 0
    np.testing.assert_allclose(x0, np.linalg.solve(A, b), rtol=0, atol=1e-10 * np.abs(x0).max())
    for _ in range(40):
        x, info = eng.dense_spd_solve(A, b)
        assert info == 0
        np.testing.assert_array_equal(x, x0)


def test_cholesky_reports_indefinite_matrix(eng):
    A = np.eye(70)
    A[50, 50] = -1.0
    _, info = eng.dense_spd_solve(A, np.ones(70))
    assert info != 0
    # several blocks: the failed pivot of block 6 must reach the end of the dataflow factorisation (NaN poisoning)
    rng = np.random.default_rng(9)
    n = 700
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    A[400, 400] = -1.0
    _, info = eng.dense_spd_solve(A, np.ones(n))
    assert info != 0
    # ... and the next factorisation on the same process is clean again
    A[400, 400] = 5.0 * n
    x, info = eng.dense_spd_solve(A, np.ones(n))
    assert info == 0
    np.testing.assert_allclose(x, np.linalg.solve(A, np.ones(n)), rtol=0, atol=1e-10 * np.abs(x).max())


@pytest.mark.parametrize("env", [{"VMM_BA_NO_DATAFLOW": "1"}, {"VMM_BA_NO_CHAIN": "1"}])
def test_cholesky_fallback_paths(eng, monkeypatch, env):
    """One k_chol_step launch per block column for ALL columns (the path of a pass that is redone after a spin gave
    up; larger systems use it for their leading columns) and the per-block back-substitution kernels, forced at sizes the dataflow kernel would otherwise handle (all of its
    workgroups resident at 1200, five times as many as compute units at 3000)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(12)
    for n in (333, 1200, 3000):
        B = rng.standard_normal((n, n))
        A = B @ B.T + n * np.eye(n)
        b = rng.standard_normal(n)
        x, info = eng.dense_spd_solve(A, b)
        assert info == 0
        ref = np.linalg.solve(A, b)
        np.testing.assert_allclose(x, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
        x2, _ = eng.dense_spd_solve(A, b)
        np.testing.assert_array_equal(x, x2)


@pytest.mark.parametrize("n", [1200, 2048, 3136])
def test_complete_panels_read_as_plain_blocks_give_the_same_bits(eng, monkeypatch, n):
    """k_chol_dataflow_bulk (VMM_BA_DF_BULK=1; the tree-ordered kernel always works this way): a workgroup that gets to a
    panel when it is complete reads the producer's compact copy instead of sweeping granules -- at 19 blocks, where every
    workgroup is resident and tracks production, at 32 and at 49 (the 34-column tail).  Same values, same order of
    operations: the solution has the same bits as with the granule-only kernel."""
    rng = np.random.default_rng(31)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    b = rng.standard_normal(n)
    monkeypatch.setenv("VMM_BA_DF_BULK", "0")
    x0, info0 = eng.dense_spd_solve(A, b)
    monkeypatch.setenv("VMM_BA_DF_BULK", "1")
    x1, info1 = eng.dense_spd_solve(A, b)
    assert info0 == 0 and info1 == 0
    np.testing.assert_array_equal(x0, x1)
    ref = np.linalg.solve(A, b)
    np.testing.assert_allclose(x1, ref, rtol=0, atol=1e-10 * np.abs(ref).max())


@pytest.mark.parametrize("n", [1408, 2048, 3072, 3136, 6000])
def test_cholesky_large_orders(eng, n):
    """22 to 48 blocks (1408, 2048, 3072): k_chol_dataflow with more workgroups than compute units (275 to 1224; only a
    prefix is resident, see dataflow_max_workgroups).  49 blocks and more (3136, 6000): one k_chol_step launch per block
    column for the leading 16 / 60 columns (rank-128 trailing updates shared by pairs of launches, tiles from an atomic
    counter, asm-requested operands, LDS ping-pong), one update-only launch, then k_chol_dataflow on the trailing 33 / 34
    block columns -- the path behind the 2000 x 1000 figures."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n // 4))
    A = B @ B.T + np.diag(rng.uniform(1.0, 2.0, n)) * n
    b = rng.standard_normal(n)
    x, info = eng.dense_spd_solve(A, b)
    assert info == 0
    ref = np.linalg.solve(A, b)
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    x2, info2 = eng.dense_spd_solve(A, b)
    assert info2 == 0
    np.testing.assert_array_equal(x, x2)


@pytest.mark.parametrize("no_xcd", ["0", "1"])
def test_syrk_many_tiles(eng, monkeypatch, no_xcd):
    """n = 4096: 528 tiles >= 512 workgroup slots, so the plan runs a full XCD-aware round of one tile per
    workgroup plus a stream-K remainder; VMM_BA_SYRK_NO_XCD=1 = the pure stream-K split."""
    monkeypatch.setenv("VMM_BA_SYRK_NO_XCD", no_xcd)
    rng = np.random.default_rng(21)
    Z = np.round(rng.standard_normal((64, 4096)) * 3.0)   # integer-valued: exact in f64, any summation order
    Cg = eng.dense_syrk(Z)
    np.testing.assert_array_equal(Cg, Z.T @ Z)
    Zr = rng.standard_normal((100, 4096))
    Cr = eng.dense_syrk(Zr)
    ref = Zr.T @ Zr
    np.testing.assert_allclose(Cr, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    np.testing.assert_array_equal(Cr, eng.dense_syrk(Zr))


def test_project_points_matches_oracle(eng, oracle, kats):
    rng = np.random.default_rng(3)
    pts = np.c_[rng.uniform(-1, 1, 50), rng.uniform(-1, 1, 50), rng.uniform(1.5, 4, 50)]
    for case in kats["obs"][:2]:
        uv = eng.project_points(case["intr"], case["dist"], pts)
        ref = np.array([oracle.project_point(case["intr"], case["dist"], p) for p in pts])
        np.testing.assert_allclose(uv, ref, rtol=0, atol=1e-9)


def test_residual_and_jacobian_kats_through_the_engine(eng, kats):
    """One camera, one tag, one observation: the accumulated blocks are J^T J of the mpmath KAT."""
    for case in kats["obs"]:
        ba = eng.BundleAdjuster(case["intr"], case["dist"], [case["cam_qt"]], [case["tag_qt"]], [case["wh"]],
                                -1, [0], [0], [case["px"]])
        r = np.array(case["residual"])
        Jc, Jt = np.array(case["J_cam"]), np.array(case["J_tag"])
        assert abs(ba.cost(robustify=False) - 0.5 * r @ r) <= 1e-12 * (0.5 * r @ r)
        blk = ba.eval_blocks(robustify=False)
        for got, ref in ((blk["V"][0], Jc.T @ Jc), (blk["U"][0], Jt.T @ Jt), (blk["W"][0], Jc.T @ Jt),
                         (blk["g_cam"][0], Jc.T @ r), (blk["g_tag"][0], Jt.T @ r)):
            np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
        ba.close()


@pytest.mark.parametrize("elim", ["cams", "tags"])
@pytest.mark.parametrize("robust", [False, True])
def test_accumulated_blocks_match_oracle(eng, oracle, elim, robust):
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(5, n_cams=7, n_tags=5)   # distortion + outliers
    mode = eng.ELIM_CAMERAS if elim == "cams" else eng.ELIM_TAGS
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam,
                            s.obs_tag, s.obs_px, elimination=mode)
    ref = _blocks_from_oracle(oracle, s, s.cam_init, s.tag_init, robust)
    got = ba.eval_blocks(robustify=robust)
    assert abs(got["cost"] - ref["cost"]) <= 1e-11 * ref["cost"]
    for k in ("V", "U", "W", "g_cam", "g_tag"):
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=1e-10 * np.abs(ref[k]).max(), err_msg=k)
    assert abs(ba.cost(robustify=robust) - ref["cost"]) <= 1e-11 * ref["cost"]
    ba.close()


def test_reprojection_statistics_match_oracle(eng, oracle):
    from visual_marker_mapping_amd.synthetic import make_scene
    s = make_scene(1, visibility=0.5)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam,
                            s.obs_tag, s.obs_px)
    sc = oracle.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
    pc, pt, avg, corner = oracle.reprojection_stats(sc)
    gc, gt, gavg, gcorner = ba.reprojection_stats()
    np.testing.assert_allclose(gcorner, corner, rtol=0, atol=1e-9)
    np.testing.assert_allclose(gc, pc, rtol=1e-12)
    np.testing.assert_allclose(gt, pt, rtol=1e-12)
    assert abs(gavg - avg) <= 1e-12 * avg
    ba.close()


def test_bad_indices_are_rejected(eng):
    from visual_marker_mapping_amd import _lib
    with pytest.raises(_lib.VmmBaError):
        eng.BundleAdjuster([1, 1, 0, 0], [0] * 5, [[1, 0, 0, 0, 0, 0, 1]], [[1, 0, 0, 0, 0, 0, 0]], [[0.1, 0.1]],
                           0, [3], [0], [[0] * 8])


def test_large_cholesky_on_the_stamps_build():
    """The diagnostic build (-DVMM_STAMPS) changes the register pressure inside the looping trailing-update workgroups
    whose loads are issued as inline asm: the same n = 2048 solve must come out right there too (a subprocess, because
    the library path is fixed at import)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "visual_marker_mapping_amd", "libvmm_ba_stamps.so")
    assert os.path.exists(lib), "build() makes it (make -C visual_marker_mapping_amd/csrc stamps)"
    code = """
import numpy as np
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(2048)
n = 2048
B = rng.standard_normal((n, n // 4))
A = B @ B.T + np.diag(rng.uniform(1.0, 2.0, n)) * n
b = rng.standard_normal(n)
x, info = eng.dense_spd_solve(A, b)
ref = np.linalg.solve(A, b)
assert info == 0 and np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max(), (info, np.abs(x - ref).max())
x2, _ = eng.dense_spd_solve(A, b)
assert np.array_equal(x, x2)
print("stamps build ok")
"""
    env = dict(os.environ, VMM_BA_LIB=lib, PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=300)
    assert out.returncode == 0 and "stamps build ok" in out.stdout, out.stdout


def test_pose_plus_matches_mpmath(eng, kats):
    """QuaternionParameterization::Plus (a5) on the device, directly against the 50-digit KATs -- until now the GPU's
    Plus was only pinned through whole trajectories."""
    qt = np.array([c["qt"] for c in kats["plus"]])
    delta = np.array([c["delta"] for c in kats["plus"]])
    ref = np.array([c["out"] for c in kats["plus"]])
    np.testing.assert_allclose(eng.pose_plus(qt, delta), ref, rtol=0, atol=4e-15)
    # a zero step is the identity, bit for bit (the LM loop relies on Plus(x, 0) == x for switched-off poses)
    np.testing.assert_array_equal(eng.pose_plus(qt, np.zeros_like(delta)), qt)
