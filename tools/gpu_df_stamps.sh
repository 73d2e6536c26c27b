# Time line of the dataflow Cholesky (diagnostic build with in-kernel stamps): per block column the start, the end of
# the consumption of the earlier panels and the end of each 8-column round of workgroup (j, j+1); phases of one round.
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
    st = (C.c_ulonglong * (32 * 64))()
    _lib.lib().vmm_ba_debug_read_df_stamps(st)
    s = np.array(list(st), dtype=np.int64).reshape(32, 64)
    t0 = s[0, 0]
    print("rep", rep, "err", np.abs(x - np.linalg.solve(A, b)).max())
    for j in range(19):
        r = (s[j, :10] - t0) / 100.0   # us
        d = (s[j, 12:14] - t0) / 100.0
        print("j=%2d start %7.2f [slice 6 seen %7.2f slice 7 seen %7.2f] consumed %7.2f rounds %s" % (j, r[0], d[0], d[1], r[1], " ".join("%7.2f" % v for v in r[2:10])))
    j = 5
    p = s[j, 40:46] - s[j, 40]
    wk = s[j, 48:55] - s[j, 48]
    print("panel 5 round J0=16 (s_memtime, unfenced: indicative): pivot wave: A-wait %d 8x8 Cholesky %d B-wait %d scale %d C-wait %d" % (p[1]-p[0], p[2]-p[1], p[3]-p[2], p[4]-p[3], p[5]-p[4]))
    print("     worker 0: phase1 %d A-wait %d phase2 %d B-wait %d phase3 %d C-wait %d" % (wk[1]-wk[0], wk[2]-wk[1], wk[3]-wk[2], wk[4]-wk[3], wk[5]-wk[4], wk[6]-wk[5]))
PY
