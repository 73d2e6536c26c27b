# C values of the paired trailing update waited for one half later (vmcnt(16) behind half 0): kernel tests, the covariance
# on every factorisation path, the factorisation time at 2000 x 1000 and 1300 x 650; eager iterations at 500 x 200
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3u
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solve.py -x -q -m gpu -k "cholesky or covariance" > gpurun_out/r3u/tests.txt 2>&1; rc=$?; tail -8 gpurun_out/r3u/tests.txt
[ $rc -eq 0 ] || exit 1
cat > /tmp/cfg4.py <<'PY'
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt) in ((1300, 650), (2000, 1000)):
    s = make_scene(4, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
    o = ba.solve(eng.default_options(robustify=0))
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
    print("%dx%d: final %.9g iters %d sync %d; cholesky %.1f us" % (nc, nt, o["final_cost"], o["num_lm_iterations"], o["num_sync_timeouts"], kt["cholesky_ms"] * 1e3), flush=True)
    ba.close()
PY
timeout -k 10 300 python /tmp/cfg4.py || exit 1
echo "--- VMM_BA_NO_DATAFLOW=1"; VMM_BA_NO_DATAFLOW=1 timeout -k 10 300 python /tmp/cfg4.py || exit 1
