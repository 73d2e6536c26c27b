# Scratch experiment: rank-k update time against the number of K slices per tile / emulated tile counts
cd $GRAFT_REPO_ROOT
for cfg in "9 0" "10 7" "10 5" "9 5" "8 0"; do
set -- $cfg
VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/libvmm_ba_exp.so VMM_BA_SYRK_SLICES=$1 VMM_BA_SYRK_DROP=$2 timeout -k 10 120 python - <<PY
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
kt = ba.time_kernels(eng.default_options(), reps=10)
print("slices", os.environ["VMM_BA_SYRK_SLICES"], "dropped tiles", os.environ["VMM_BA_SYRK_DROP"], "syrk+reduce us", round(kt["syrk_ms"] * 1e3, 1))
ba.close()
PY
done
