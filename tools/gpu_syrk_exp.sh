# Rank-k update (+ partial-tile sum) time against the number of K slices per tile (VMM_BA_SYRK_SLICES overrides the
# slot-count rule of make_syrk_plan).  The time break-down and the 48-tile emulation quoted in DESIGN.md 4.2 / 8 came
# from throw-away builds of this kernel with switches (skip the stores, skip the MFMAs, launch part of the items,
# drop tiles); those switches are not in the tree.
cd $GRAFT_REPO_ROOT
for x in 8 9 10; do
VMM_BA_SYRK_SLICES=$x timeout -k 10 120 python - <<PY
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
kt = ba.time_kernels(eng.default_options(), reps=10)
print("slices", os.environ["VMM_BA_SYRK_SLICES"], "syrk + partial-tile sum (us)", round(kt["syrk_ms"] * 1e3, 1),
      "iteration (us)", round(kt["lm_iteration_ms"] * 1e3, 1))
ba.close()
PY
done
