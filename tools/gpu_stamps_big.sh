cd $GRAFT_REPO_ROOT
timeout -k 10 500 python - <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace('libvmm_ba.so', 'libvmm_ba_stamps.so')
from visual_marker_mapping_amd import engine as eng
rng = np.random.default_rng(0)
n = 6000
B = rng.standard_normal((n, 200)); A = B @ B.T + n * np.eye(n); b = rng.standard_normal(n)
for rep in range(2):
    x, info = eng.dense_spd_solve(A, b)
    st = (C.c_ulonglong * 64)()
    _lib.lib().vmm_ba_debug_read_stamps(st, 64)
    s = list(st)
    u = s[40:47]
    print("update tile (2nd of WG 0, k=1): issue loads %d | MFMA loop %d | wait loads %d | stores %d | LDS park %d | barrier %d | total %d cycles" % (u[1]-u[0], u[2]-u[1], u[3]-u[2], u[4]-u[3], u[5]-u[4], u[6]-u[5], u[6]-u[0]))
PY
