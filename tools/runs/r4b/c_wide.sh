cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "syrk" > $O/pytest_syrk.txt 2>&1; tail -5 $O/pytest_syrk.txt
timeout -k 10 500 python - > $O/wide.txt 2>&1 <<'PY'
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
for rep in range(2):
    for w in ("0", "1"):
        os.environ["VMM_BA_SYRK_WIDE"] = w
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        sm = ba.solve(eng.default_options(robustify=0))
        kt = ba.time_kernels(reps=30)
        print("wide", w, "syrk+reduce us %.1f" % (kt["syrk_ms"] * 1e3), "iterations", sm["iterations"] if isinstance(sm, dict) else sm, flush=True)
        del ba
PY
cat $O/wide.txt
for w in 0 1; do VMM_BA_SYRK_WIDE=$w timeout -k 10 300 python bench.py --no-cpu-baseline --steps 70 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('wide $w', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()})"; done
