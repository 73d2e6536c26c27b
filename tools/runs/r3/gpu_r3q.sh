# k_chol_step + dataflow tail: kernel tests, then the factorisation time at 2000 x 1000 against the tail length.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cholesky" > gpurun_out/r3q/tests.txt 2>&1; rc=$?; tail -15 gpurun_out/r3q/tests.txt
[ $rc -eq 0 ] || exit 1
cat > /tmp/cfg4.py <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(4)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
o = ba.solve(eng.default_options(robustify=0))
kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
print("final %.9g iters %d sync %d; cholesky %.1f us" % (o["final_cost"], o["num_lm_iterations"], o["num_sync_timeouts"], kt["cholesky_ms"] * 1e3), flush=True)
ba.close()
PY
for t in 0 24 30 34 38 44; do echo "--- tail $t"; VMM_BA_CHOL_TAIL=$t timeout -k 10 300 python /tmp/cfg4.py || exit 1; done
