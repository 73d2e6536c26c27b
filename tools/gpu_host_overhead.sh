# Host-side cost around a solve (the bench runs 7-iteration solves back to back): set_state, solve wall time, device phases.
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python - <<'PY'
import time
import numpy as np
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
o = eng.default_options(robustify=0)
for _ in range(5):
    ba.set_state(s.cam_init, s.tag_init); ba.solve(o)
ts, tv, dev, lib = [], [], [], []
for _ in range(50):
    t0 = time.perf_counter(); ba.set_state(s.cam_init, s.tag_init); t1 = time.perf_counter()
    out = ba.solve(o); t2 = time.perf_counter()
    ts.append(t1 - t0); tv.append(t2 - t1); lib.append(out["time_solve_s"])
    dev.append(sum(out[k] for k in ("time_eval_s", "time_eliminate_s", "time_factor_solve_s", "time_step_s", "time_control_s")))
print("per solve (us): set_state %.1f  solve (python wall) %.1f  solve (inside the library) %.1f  device phases %.1f  -> host overhead %.1f of %.1f"
      % (1e6 * np.median(ts), 1e6 * np.median(tv), 1e6 * np.median(lib), 1e6 * np.median(dev),
         1e6 * (np.median(ts) + np.median(tv) - np.median(dev)), 1e6 * (np.median(ts) + np.median(tv))))
ba.close()
PY
