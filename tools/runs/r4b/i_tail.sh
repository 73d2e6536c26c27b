cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_i; mkdir -p $O
timeout -k 10 800 python - > $O/tail.txt 2>&1 <<'PY'
import os
for tail in (34, 40, 46, 52, 58, 28):
    os.environ["VMM_BA_CHOL_TAIL"] = str(tail)
    # VMM_BA_CHOL_TAIL is read once per process (static): run each setting in a child process
    import subprocess, sys, json
    code = r'''
import os, numpy as np, time
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(4)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
o = eng.default_options(robustify=0); o.max_num_iterations = 2
ba.solve(o)
kt = ba.time_kernels(eng.default_options(robustify=0), reps=5)
print("tail", os.environ["VMM_BA_CHOL_TAIL"], "cholesky us %.1f" % (kt["cholesky_ms"] * 1e3), flush=True)
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-400:], flush=True)
PY
cat $O/tail.txt
