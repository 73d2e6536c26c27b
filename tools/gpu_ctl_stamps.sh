# Where the control kernel's time goes (diagnostic build with in-kernel s_memrealtime stamps, 100 MHz).
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python - <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libvmm_ba.so', 'libvmm_ba_stamps.so')
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(2)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
for it in (2, 3, 4):
    ba.set_state(s.cam_init, s.tag_init)
    o = ba.solve(eng.default_options(robustify=0, max_num_iterations=it))
    st = (C.c_ulonglong * 16)()
    _lib.lib().vmm_ba_debug_read_ctl_stamps(st)
    t = np.array(list(st)[:7], dtype=np.int64)
    d = (t[1:] - t[:-1]) / 100.0
    print("last k_control of a %d-iteration solve (us): loads+per-pose work %.2f  block reduce %.2f  thread 0 logic %.2f  store ctl %.2f  barrier %.2f  copies + LM diagonal %.2f  | total %.2f" % (it, d[0], d[1], d[2], d[3], d[4], d[5], (t[6] - t[0]) / 100.0))
ba.close()
PY
