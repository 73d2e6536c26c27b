# Scratch experiment: rank-k update time against the number of K slices per tile
cd $GRAFT_REPO_ROOT
for x in 8 9; do
VMM_BA_SYRK_SLICES=$x timeout -k 10 120 python - <<PY
import os
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
kt = ba.time_kernels(eng.default_options(), reps=10)
o = ba.solve(eng.default_options(), trace_capacity=32)
print("slices", os.environ["VMM_BA_SYRK_SLICES"], {k: round(v * 1e3, 1) for k, v in kt.items()}, o["iterations"], o["final_cost"])
ba.close()
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k syrk 2>&1 | tail -2
