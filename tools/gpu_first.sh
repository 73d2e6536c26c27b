set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|^CPU\(s\)" ; rocminfo | grep -E "Marketing|gfx" | head -4
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, 0, s.obs_cam, s.obs_tag, s.obs_px)
print(ba.time_kernels(eng.default_options(robustify=0), reps=5))
for poll in (1, 8):
    ba.set_state(s.cam_init, s.tag_init)
    t = time.time(); out = ba.solve(eng.default_options(robustify=0, poll_interval=poll), trace_capacity=32); dt = time.time() - t
    print(poll, dt, out['iterations'], out['num_lm_iterations'], out['time_solve_s'], out['final_cost'])
for t in out['trace']: print(t)
PY
