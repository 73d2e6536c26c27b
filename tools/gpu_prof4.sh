# per-launch durations of k_chol_step on the 2000 x 1000 problem
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof4
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof4 -o p4 -- python bench.py --config 4 --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/prof4/bench.json 2> gpurun_out/prof4/err.log || exit 1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof4/*kernel_trace.csv")[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))
rows.sort()
# one full Cholesky: the window of 94 consecutive k_chol_step launches with the largest total (launches behind the end
# of an LM loop return at once)
idx = [i for i, r in enumerate(rows) if "k_chol_step" in r[2]]
best, seq = -1.0, idx[:94]
for w in range(0, len(idx) - 93, 94):
    tot = sum(rows[i][1] - rows[i][0] for i in idx[w:w + 94])
    if tot > best:
        best, seq = tot, idx[w:w + 94]
out = open("gpurun_out/prof4/chol_steps.txt", "w")
tot = 0
for n, i in enumerate(seq):
    d = (rows[i][1] - rows[i][0]) / 1e3
    gap = (rows[i][0] - rows[i - 1][1]) / 1e3
    tot += d
    out.write("k=%d grid=%s dur_us=%.1f gap_us=%.1f\n" % (n, rows[i][3], d, gap))
out.write("total_us=%.1f\n" % tot)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    agg[r[2]][0] += 1; agg[r[2]][1] += (r[1] - r[0]) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    out.write("%-60s calls=%d total_us=%.0f avg_us=%.1f\n" % (k[:60], v[0], v[1], v[1] / v[0]))
PY
find gpurun_out/prof4 -name '*.csv' -size +2M -delete
cat gpurun_out/prof4/chol_steps.txt | awk 'NR%6==1 || /total|calls/'
