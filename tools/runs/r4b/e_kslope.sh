cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_e; mkdir -p $O
timeout -k 10 500 python - > $O/kslope.txt 2>&1 <<'PY'
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba0 = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
ba0.solve(eng.default_options(robustify=0))
cams, tags = ba0.get_state()
for w in ("1", "0"):
    os.environ["VMM_BA_SYRK_WIDE"] = w
    for kt in (188, 140, 94, 48, 20, 10):
        os.environ["VMM_BA_DEBUG_SYRK_KT"] = str(kt)
        ba = eng.BundleAdjuster(s.intr, s.dist, cams, tags, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
        kt_ = ba.time_kernels(reps=30)
        print("wide", w, "stages", kt, "syrk+reduce us %.1f" % (kt_["syrk_ms"] * 1e3), flush=True)
        del ba
PY
cat $O/kslope.txt
