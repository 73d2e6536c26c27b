cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_t; mkdir -p $O
timeout -k 10 300 python - > $O/bits.txt 2>&1 <<'PY'
import os, numpy as np
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(41)
for n in (700, 1200, 2048, 3136):
    B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
    os.environ["VMM_BA_DF_HELP"] = "0"; x0, i0 = eng.dense_spd_solve(A, b)
    os.environ["VMM_BA_DF_HELP"] = "1"; x1, i1 = eng.dense_spd_solve(A, b)
    ref = np.linalg.solve(A, b)
    print(n, "info", i0, i1, "same bits", bool(np.array_equal(x0, x1)), "err", float(np.abs(x1 - ref).max() / np.abs(ref).max()), flush=True)
PY
cat $O/bits.txt; grep -q "same bits False\|Traceback" $O/bits.txt && exit 1
VMM_BA_DF_HELP=1 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py -m gpu -x -q -k "not syrk" > $O/pytest_help.txt 2>&1; rc=$?; tail -4 $O/pytest_help.txt; [ $rc -eq 0 ] || exit 1
b() { name=$1; shift; "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(d['value'],1), round(d.get('ms_per_step', 0),4), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items() if k in ('cholesky_solve',)})"; }
for h in 0 1; do
export VMM_BA_DF_HELP=$h
b "headline help=$h" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 140
b "closeup help=$h" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b "corridor help=$h" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --wall-rows 2 --steps 70
b "closeup2000 help=$h" timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
b "cfg4 help=$h" timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline
done
unset VMM_BA_DF_HELP
timeout -k 10 400 python - <<'PY'
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt) in ((400, 250), (600, 320), (800, 400), (1000, 500)):
    s = make_scene(2, n_cams=nc, n_tags=nt)
    for w in ("0", "1"):
        os.environ["VMM_BA_DF_HELP"] = w
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        o = ba.solve(eng.default_options(robustify=0))
        kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
        print(nc, nt, "help", w, "cholesky us %.1f" % (kt["cholesky_ms"] * 1e3), "iteration us %.1f" % (kt["lm_iteration_ms"] * 1e3), "cost %.6f" % o["final_cost"], "timeouts", o["num_sync_timeouts"], flush=True)
        ba.close()
PY
