cd $GRAFT_REPO_ROOT
timeout -k 10 300 python - <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libvmm_ba.so', 'libvmm_ba_stamps.so')
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(0)
n = 1200
B = rng.standard_normal((n, n)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
for rep in range(4):
    x, info = eng.dense_spd_solve(A, b)
    st = (C.c_ulonglong * 32)()
    _lib.lib().vmm_ba_debug_read_stamps(st, 32)
    s = list(st)
    print("round 32: publish %d bar %d chol4+x %d bar %d mfma-issue %d" % (s[7]-s[6], s[8]-s[7], s[9]-s[8], s[10]-s[9], s[11]-s[10])); print("real100MHz total", s[21]-s[16]); print("S-load %d pre-update %d rounds %d store %d total %d cycles" % (s[6]-s[0], s[1]-s[6], s[2]-s[1], s[5]-s[4], s[5]-s[0]))
    print("err", np.abs(x - np.linalg.solve(A, b)).max())
PY
