# Experiment: would a rank-128 trailing update pay?  Variant K2 runs the MFMA/LDS work of two rank-64 updates per visit of
# a C tile (same operands twice, result halved): per-step time against the normal build at 2000 x 1000.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cat > /tmp/cfg4.py <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(4)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
o = ba.solve(eng.default_options(robustify=0))
kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
print("final %.9g; cholesky %.1f us" % (o["final_cost"], kt["cholesky_ms"] * 1e3), flush=True)
ba.close()
PY
cat > /tmp/steps.py <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r3o/**/%s_kernel_trace.csv" % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_chol_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-94:]]
print(sys.argv[1], "last factorisation: steps 1,5,10,20,30,40,50,60:", [round(d[k], 1) for k in (1, 5, 10, 20, 30, 40, 50, 60)], "total", round(sum(d), 1))
PY
echo "--- normal"; timeout -k 10 300 python /tmp/cfg4.py
echo "--- K2"; VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/libvmm_ba_K2.so timeout -k 10 300 python /tmp/cfg4.py
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3o -o k2a -- python3 /tmp/cfg4.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python /tmp/steps.py k2a
find gpurun_out/r3o -name '*.csv' -size +4M -delete
cd /tmp && VMM_BA_LIB=$GRAFT_REPO_ROOT/visual_marker_mapping_amd/libvmm_ba_K2.so timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3o -o k2b -- python3 /tmp/cfg4.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python /tmp/steps.py k2b
find gpurun_out/r3o -name '*.csv' -size +4M -delete
