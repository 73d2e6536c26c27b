# Experiment: how far does the one-launch dataflow factorisation stay ahead of k_chol_step?
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
cat > /tmp/mid.py <<'PY'
import sys, time, numpy as np
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(5)
for n in (1536, 2432, 3200, 4096, 6000):
    B = rng.standard_normal((n, n // 4))
    A = B @ B.T + np.diag(rng.uniform(1.0, 2.0, n)) * n
    b = rng.standard_normal(n)
    ref = np.linalg.solve(A, b)
    x, info = eng.dense_spd_solve(A, b)
    t0 = time.perf_counter()
    for _ in range(3):
        x, info = eng.dense_spd_solve(A, b)
    dt = (time.perf_counter() - t0) / 3
    print("n=%d blocks %d info %d err %.2e  (%.1f ms per call incl. upload)" % (n, (n + 63) // 64, info, np.abs(x - ref).max() / np.abs(ref).max(), dt * 1e3), flush=True)
PY
cat > /tmp/cfg4.py <<'PY'
import time
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt) in ((1000, 500), (1200, 640), (2000, 1000)):
    s = make_scene(4, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
    o = ba.solve(eng.default_options(robustify=0))
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
    print("%dx%d blocks %d: %d iters final %.9g sync %d/%d; cholesky %.1f us syrk %.1f us iter %.1f us" % (
        nc, nt, (6 * nt + 63) // 64, o["num_lm_iterations"], o["final_cost"], o["num_sync_timeouts"], o["sync_timeout_kernels"],
        kt["cholesky_ms"] * 1e3, kt["syrk_ms"] * 1e3, kt["lm_iteration_ms"] * 1e3), flush=True)
    ba.close()
PY
export PYTHONPATH=$GRAFT_REPO_ROOT
echo "--- default"; timeout -k 10 400 python /tmp/cfg4.py
echo "--- VMM_BA_DF_MAX_WG=100000"; VMM_BA_DF_MAX_WG=100000 timeout -k 10 400 python /tmp/cfg4.py
echo "--- dense solves, VMM_BA_DF_MAX_WG=100000"; VMM_BA_DF_MAX_WG=100000 timeout -k 10 400 python /tmp/mid.py
