# Where the CPU oracle's time goes on this host (cpu_baseline context): evaluation vs linear solve, by thread count.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.')
from oracle import oracle as O
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
for th in (min(os.cpu_count(), 64), 16, 1):
    sc = O.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    O.solve(sc, O.default_options(robustify=0, num_threads=th, max_num_iterations=1))
    sc = O.Scene(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
    t0 = time.perf_counter()
    summ, trace = O.solve(sc, O.default_options(robustify=0, num_threads=th, max_num_iterations=3 if th == 1 else 50))
    dt = time.perf_counter() - t0
    n = max(summ["num_cost_evals"], 1)
    print("%2d threads: %d LM iterations in %.2f s = %.1f ms each; evaluation %.1f ms, linear solve %.1f ms per iteration"
          % (th, n, dt, dt / n * 1e3, summ["time_eval_s"] / n * 1e3, summ["time_linear_s"] / n * 1e3))
PY
