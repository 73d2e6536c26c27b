# where does the incremental reconstruction's wall time go? (cProfile of the resident-handle run)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
export PYTHONPATH=$GRAFT_REPO_ROOT
cat > /tmp/prof_incr.py <<'PY'
import cProfile, pstats, io, contextlib, time
from visual_marker_mapping_amd.synthetic import make_scene
from visual_marker_mapping_amd.tag_reconstructor import CameraModel, TagReconstructor, detection_result_from_arrays
s = make_scene(2, n_cams=100, n_tags=60, visibility=0.3)
def run():
    det = detection_result_from_arrays(s.obs_cam, s.obs_tag, s.obs_px, s.tag_wh, 100)
    rec = TagReconstructor(det)
    rec.setCameraModel(CameraModel(*[float(v) for v in s.intr], s.dist, 4000, 6000))
    with contextlib.redirect_stdout(io.StringIO()):
        rec.startReconstruction(1, deviceResident=True)
    rec.close()
run()
t0 = time.perf_counter(); run(); print("wall %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(32); print(out.getvalue()[:6000])
PY
timeout -k 10 300 python /tmp/prof_incr.py > gpurun_out/r3v/profile.txt 2>&1; tail -60 gpurun_out/r3v/profile.txt
