# k_chol_step with rank-128 pair updates: kernel tests, then the per-launch durations and the factorisation time at
# 2000 x 1000 (and the mid sizes that take this path when VMM_BA_NO_DATAFLOW=1).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cholesky" > gpurun_out/r3p/tests.txt 2>&1; rc=$?; tail -15 gpurun_out/r3p/tests.txt
[ $rc -eq 0 ] || exit 1
cat > /tmp/cfg4.py <<'PY'
from visual_marker_mapping_amd import engine as eng
from visual_marker_mapping_amd.synthetic import make_scene
s = make_scene(4)
ba = eng.BundleAdjuster(s.intr, s.dist, s.cam_init, s.tag_init, s.tag_wh, s.fixed_tag, s.obs_cam, s.obs_tag, s.obs_px, precision=eng.PRECISION_F32_ACCUM)
o = ba.solve(eng.default_options(robustify=0))
kt = ba.time_kernels(eng.default_options(robustify=0), reps=3)
print("final %.9g iters %d; cholesky %.1f us" % (o["final_cost"], o["num_lm_iterations"], kt["cholesky_ms"] * 1e3), flush=True)
ba.close()
PY
cat > /tmp/steps.py <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r3p/**/%s_kernel_trace.csv" % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_chol_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-94:]]
print(sys.argv[1], "last factorisation, per launch:", [round(x, 1) for x in d], "total", round(sum(d), 1))
PY
timeout -k 10 300 python /tmp/cfg4.py || exit 1
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3p -o pair -- python3 /tmp/cfg4.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python /tmp/steps.py pair > gpurun_out/r3p/steps.txt; cut -c1-1200 gpurun_out/r3p/steps.txt
find gpurun_out/r3p -name '*.csv' -size +4M -delete
