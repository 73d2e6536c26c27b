# Time line of the tree-ordered dataflow Cholesky on the close-up scene (diagnostic build with in-kernel stamps): per block
# column the start, the end of the consumption of the panels it depends on and the end of each 8-column round, next to the
# block structure of the factor (VMM_BA_DEBUG=1).
cd $GRAFT_REPO_ROOT
export VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/libvmm_ba_stamps.so VMM_BA_DEBUG=1
timeout -k 10 300 python - "$@" <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from visual_marker_mapping_amd import _lib
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
kw = dict(neighbors_min=6, neighbors_max=10)
if len(sys.argv) > 1:
    kw["wall_rows"] = int(sys.argv[1])
s = make_scene(2, **kw)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px)
for rep in range(3):
    ba.set_state(s.cam_init, s.tag_init)
    out = ba.solve(eng.default_options(max_num_iterations=1))
st = (C.c_ulonglong * (32 * 128))()
_lib.lib().vmm_ba_debug_read_df_stamps(st)
v = np.array(list(st), dtype=np.int64).reshape(32, 128)
nb = out["tree_ordering"]
t0 = v[:, 0][v[:, 0] > 0].min()
print("tree nodes", nb, "block_sparse", out["block_sparse"])
for j in range(32):
    if v[j, 0] <= 0 or v[j, 9] <= 0:
        continue
    r = (v[j, :10] - t0) / 100.0
    print("j=%2d start %7.2f consumed %7.2f rounds %s" % (j, r[0], r[1], " ".join("%7.2f" % x for x in r[2:10])))
ba.close()
PY
