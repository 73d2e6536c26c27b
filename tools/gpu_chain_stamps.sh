# Time line of the chained back-substitution (diagnostic build with in-kernel s_memrealtime stamps, 100 MHz).
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
for rep in range(3):
    x, info = eng.dense_spd_solve(A, b)
    st = (C.c_ulonglong * (128 * 4))()
    _lib.lib().vmm_ba_debug_read_chain_stamps(st)
    s = np.array(list(st), dtype=np.int64).reshape(128, 4)[:19]
    t0 = s[:, 0].min()
    print("rep", rep, "err", np.abs(x - np.linalg.solve(A, b)).max())
    for m in range(18, -1, -1):
        print("block %2d: started %6.2f  last dependency seen %6.2f  published %6.2f   (work %5.2f us, hand-off from block %d: %5.2f us)"
              % (m, (s[m, 0] - t0) / 100.0, (s[m, 1] - t0) / 100.0, (s[m, 2] - t0) / 100.0, (s[m, 2] - s[m, 1]) / 100.0,
                 m + 1, ((s[m, 1] - s[m + 1, 2]) / 100.0) if m < 18 else 0.0))
PY
