# mixed paired / single k_chol_step schedule: kernel tests, 2000 x 1000, mid sizes on the step path
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3r
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cholesky" > gpurun_out/r3r/tests.txt 2>&1; rc=$?; tail -15 gpurun_out/r3r/tests.txt
[ $rc -eq 0 ] || exit 1
cat > /tmp/cfg4.py <<'PY'
import sys
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt, prec) in ((400, 250, 0), (1000, 500, 0), (1300, 650, 1), (2000, 1000, 1)):
    s = make_scene(4 if prec else 2, n_cams=nc, n_tags=nt)
    ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM if prec else eng.PRECISION_F64)
    o = ba.solve(eng.default_options(robustify=0))
    kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
    print("%dx%d blocks %d final %.9g iters %d sync %d; cholesky %.1f us" % (nc, nt, (6 * nt + 63) // 64, o["final_cost"], o["num_lm_iterations"], o["num_sync_timeouts"], kt["cholesky_ms"] * 1e3), flush=True)
    ba.close()
PY
timeout -k 10 400 python /tmp/cfg4.py || exit 1
echo "--- VMM_BA_NO_DATAFLOW=1"
VMM_BA_NO_DATAFLOW=1 timeout -k 10 400 python /tmp/cfg4.py || exit 1
