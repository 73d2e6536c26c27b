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
    st = (C.c_ulonglong * (32 * 128))()
    _lib.lib().vmm_ba_debug_read_df_stamps(st)
    s = np.array(list(st), dtype=np.int64).reshape(32, 128)
    t0 = s[0, 0]
    print("rep", rep, "err", np.abs(x - np.linalg.solve(A, b)).max())
    for j in range(19):
        r = (s[j, :10] - t0) / 100.0   # us
        d = (s[j, 12:14] - t0) / 100.0
        print("j=%2d start %7.2f [slice 6 seen %7.2f slice 7 seen %7.2f] consumed %7.2f rounds %s" % (j, r[0], d[0], d[1], r[1], " ".join("%7.2f" % v for v in r[2:10])))
    for j in (5, 6):
        print("j=%d: slices of panel %d published at %s" % (j, j - 1, " ".join("%7.2f" % v for v in (s[j - 1, 2:10] - t0) / 100.0)))
        print("      seen by workgroup (%d, %d) at         %s (even: its worker wave 0, odd: its pivot wave 0)" % (j, j + 1, " ".join("%7.2f" % v for v in (s[j, 14:22] - t0) / 100.0)))
    j = 5
    p = s[j, 40:45] - s[j, 40]
    wk = s[j, 48:53] - s[j, 48]
    print("panel 5 round J0=16 (s_memtime, unfenced: indicative): pivot wave: 8x8 Cholesky %d B-wait %d scale + next pivot block %d C-wait %d" % (p[1]-p[0], p[2]-p[1], p[3]-p[2], p[4]-p[3]))
    p1 = s[j, 32:37] - s[j, 32]
    w1 = s[j, 24:29] - s[j, 24]
    print("     pivot wave 1: 8x8 Cholesky %d B-wait %d scale + publish %d C-wait %d" % (p1[1]-p1[0], p1[2]-p1[1], p1[3]-p1[2], p1[4]-p1[3]))
    print("     worker 1: phase 2 %d B-wait %d phase 3 %d C-wait %d" % (w1[1]-w1[0], w1[2]-w1[1], w1[3]-w1[2], w1[4]-w1[3]))
    print("     raw s_memtime relative to pivot wave 0's round start: P0 %s | P1 %s | W0 %s | W1 %s" % (s[j, 40:45] - s[j, 40], s[j, 32:37] - s[j, 40], s[j, 48:53] - s[j, 40], s[j, 24:29] - s[j, 40]))
    q = s[j, 56:59] - s[j, 42]
    print("     after B: scaled %d, rows written + Nd read %d, Gram done %d" % (q[0], q[1], q[2]))
    print("     worker 0: phase 2 %d B-wait %d phase 3 %d C-wait %d" % (wk[1]-wk[0], wk[2]-wk[1], wk[3]-wk[2], wk[4]-wk[3]))
PY
