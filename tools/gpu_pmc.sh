# HBM traffic counters, one --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit in one pass).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc -o pmc_$c -- python bench.py --steps 14 --warmup 7 --no-cpu-baseline > gpurun_out/pmc/bench_$c.json 2> gpurun_out/pmc/err_$c.log || exit 1
done
ls -la gpurun_out/pmc | head -20
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc/pmc_%s_counter_collection.csv" % c)
    if not f:
        print("no counter file for", c, glob.glob("gpurun_out/pmc/*")); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmc/pmc_summary.json", "w"), indent=1)
for c, d in out.items():
    print(c)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print("  %-60s %12.1f per dispatch (%d)" % (k[:60], v["sum"] / max(v["dispatches"], 1), v["dispatches"]))
PY
find gpurun_out/pmc -name '*.csv' -size +8M -delete
