cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_b; mkdir -p $O
timeout -k 10 500 python - > $O/desync.txt 2>&1 <<'PY'
import os, time
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
for rep in range(2):
    for d in (0, 1, 2, 4, 6, 8, 12, 16, 24):
        os.environ["VMM_BA_SYRK_DESYNC"] = str(d)
        ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        ba.solve(eng.default_options(robustify=0))
        kt = ba.time_kernels(reps=30)
        print("desync", d, "syrk+reduce us %.1f" % (kt["syrk_ms"] * 1e3), "iteration us %.1f" % (kt["lm_iteration_ms"] * 1e3), flush=True)
        del ba
PY
cat $O/desync.txt
