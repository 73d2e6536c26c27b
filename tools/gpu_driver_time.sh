# wall time of the incremental driver (startReconstruction) on a 60 x 40 scene: device-resident handle + masks,
# rebuilt handles (with reuse between a solve and its statistics), rebuilt handles without reuse
cd $GRAFT_REPO_ROOT
for mode in resident cold cold_nocache; do
VMM_DRIVER_MODE=$mode VMM_BA_NO_HANDLE_CACHE=$( [ $mode = cold_nocache ] && echo 1 ) timeout -k 10 500 python - <<'PY'
import os, sys, time, io, contextlib
sys.path.insert(0, '.')
from visual_marker_mapping_amd import synthetic
from visual_marker_mapping_amd.tag_reconstructor import TagReconstructor, CameraModel, detection_result_from_arrays
s = synthetic.make_scene(1, n_cams=60, n_tags=40, visibility=0.5)
det = detection_result_from_arrays(s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, 60)
rec = TagReconstructor(det)
rec.setCameraModel(CameraModel(*s.intr, s.dist, 4000, 6000))
buf = io.StringIO()
mode = os.environ["VMM_DRIVER_MODE"]
t0 = time.time()
with contextlib.redirect_stdout(buf):
    rec.startReconstruction(1, deviceResident=(mode == "resident"))
dt = time.time() - t0
print("%-13s startReconstruction 60 x 40: %.2f s, %d cameras, %d tags, %d BA solves" % (
    mode, dt, len(rec.reconstructedCameras), len(rec.reconstructedTags), buf.getvalue().count("Solution ")))
PY
done
