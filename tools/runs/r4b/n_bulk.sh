cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chol or cholesky or spd or complete_panels" > $O/pytest_chol.txt 2>&1; rc=$?; tail -4 $O/pytest_chol.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sync_timeout.py -m gpu -x -q > $O/pytest_tree.txt 2>&1; rc=$?; tail -4 $O/pytest_tree.txt; [ $rc -eq 0 ] || exit 1
b() { name=$1; shift; "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(d['value'],1), round(d.get('ms_per_step', 0),4), {k:round(v['ms']*1000,1) for k,v in d.get('kernels', {}).items() if k in ('cholesky_solve','schur_syrk')})"; }
for bulk in 0 1; do
export VMM_BA_DF_BULK=$bulk
b "closeup bulk=$bulk" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --steps 70
b "corridor bulk=$bulk" timeout -k 10 300 python bench.py --no-cpu-baseline --neighbors 6 10 --wall-rows 2 --steps 70
b "closeup2000 bulk=$bulk" timeout -k 10 400 python bench.py --no-cpu-baseline --config 4 --neighbors 6 10 --precision f64 --steps 30 --warmup 10
b "cfg4 bulk=$bulk" timeout -k 10 400 python bench.py --config 4 --steps 24 --warmup 8 --no-cpu-baseline
done
unset VMM_BA_DF_BULK
timeout -k 10 400 python - <<'PY'
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
for (nc, nt) in ((400, 250), (600, 320), (800, 400), (1000, 500)):
    s = make_scene(2, n_cams=nc, n_tags=nt)
    for w in ("0", "1"):
        os.environ["VMM_BA_DF_BULK"] = w
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        o = ba.solve(eng.default_options(robustify=0))
        kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
        print(nc, nt, "bulk", w, "cholesky us %.1f" % (kt["cholesky_ms"] * 1e3), "iteration us %.1f" % (kt["lm_iteration_ms"] * 1e3), "cost %.6f" % o["final_cost"], "timeouts", o["num_sync_timeouts"], flush=True)
        ba.close()
PY
