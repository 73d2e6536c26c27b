cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r4b_k; mkdir -p $O
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4b_k/**/prof_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_chol_step" in r["Kernel_Name"]]
# one factorisation = 61 launches (60 steps + hand-over); print the last complete one
n = 61
last = steps[-n:]
print("launches of the last factorisation (us):", " ".join("%.0f" % x for x in last))
print("sum %.1f us" % sum(last))
PY
find gpurun_out/r4b_k -name '*.csv' -size +2M -delete
