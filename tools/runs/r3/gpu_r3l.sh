# Experiment: the one-launch dataflow factorisation with more workgroups than compute units (mid-size reduced systems).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
cat > /tmp/mid.py <<'PY'
import sys, time, numpy as np
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt) in ((400, 250), (600, 320), (800, 400)):
    s = make_scene(2, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    o = ba.solve(eng.default_options(robustify=0), trace_capacity=64)
    ba.set_state(s.cam_init, s.tag_init)
    t0 = time.perf_counter(); o = ba.solve(eng.default_options(robustify=0), trace_capacity=64); dt = time.perf_counter() - t0
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
    print("%dx%d blocks %d: %d iters %.2f ms/solve, final %.9g, sync %d/%d; cholesky %.1f us syrk %.1f us iter %.1f us" % (
        nc, nt, (6 * nt + 63) // 64, o["num_lm_iterations"], dt * 1e3, o["final_cost"], o["num_sync_timeouts"], o["sync_timeout_kernels"],
        kt["cholesky_ms"] * 1e3, kt["syrk_ms"] * 1e3, kt["lm_iteration_ms"] * 1e3))
    ba.close()
PY
echo "--- default (<= 256 workgroups: k_chol_step for these sizes)"; PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python /tmp/mid.py
echo "--- VMM_BA_DF_MAX_WG=1300"; VMM_BA_DF_MAX_WG=1300 PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python /tmp/mid.py
