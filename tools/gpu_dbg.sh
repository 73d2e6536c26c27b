cd $GRAFT_REPO_ROOT
VMM_BA_DEBUG=1 timeout -k 10 300 python - <<'PY'
import sys
sys.path.insert(0, '.')
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
print(ba.time_kernels(eng.default_options(robustify=0), reps=3))
PY
