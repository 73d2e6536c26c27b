# the incremental driver on a 200 x 100 scene (25 % visibility), device-resident
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python - <<'PY'
import sys, time, io, contextlib
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import synthetic
from visual_marker_mapping_amd.tag_reconstructor import TagReconstructor, CameraModel, detection_result_from_arrays
s = synthetic.make_scene(5, n_cams=200, n_tags=100, visibility=0.25)
det = detection_result_from_arrays(s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, 200)
rec = TagReconstructor(det)
rec.setCameraModel(CameraModel(*s.intr, s.dist, 4000, 6000))
buf = io.StringIO()
t0 = time.time()
with contextlib.redirect_stdout(buf):
    rec.startReconstruction(1)
dt = time.time() - t0
err = [np.abs(rec.reconstructedTags[t].t - s.tag_gt[t, 4:]).max() for t in rec.reconstructedTags]
out = buf.getvalue()
print("200 x 100 (%d observations, distortion + 2%% outliers): %.1f s, %d cameras, %d tags, %d BA solves, "
      "%d tags / %d cameras pruned, max tag position error %.2e m" % (
          s.n_obs, dt, len(rec.reconstructedCameras), len(rec.reconstructedTags), out.count("Solution "),
          out.count("Removing bad marker"), out.count("Removing bad camera"), max(err)))
print(out.strip().splitlines()[-1])
PY
