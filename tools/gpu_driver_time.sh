# wall time of the incremental driver (startReconstruction) on a 60 x 40 scene, with and without handle reuse
cd $GRAFT_REPO_ROOT
for nc in 0 1; do
VMM_BA_NO_HANDLE_CACHE=$( [ $nc = 1 ] && echo 1 ) timeout -k 10 500 python - <<'PY'
import os, sys, time, io, contextlib
sys.path.insert(0, '.')
from visual_marker_mapping_amd import synthetic
from visual_marker_mapping_amd.tag_reconstructor import TagReconstructor, CameraModel, detection_result_from_arrays
s = synthetic.make_scene(1, n_cams=60, n_tags=40, visibility=0.5)
det = detection_result_from_arrays(s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, 60)
rec = TagReconstructor(det)
rec.setCameraModel(CameraModel(*s.intr, s.dist, 4000, 6000))
buf = io.StringIO()
t0 = time.time()
with contextlib.redirect_stdout(buf):
    rec.startReconstruction(1)
dt = time.time() - t0
print("handle cache %s: startReconstruction 60 x 40: %.2f s, %d cameras, %d tags, %d BA solves" % (
    "off" if os.environ.get("VMM_BA_NO_HANDLE_CACHE") else "on", dt, len(rec.reconstructedCameras),
    len(rec.reconstructedTags), buf.getvalue().count("Solution ")))
PY
done
