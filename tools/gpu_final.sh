# Round-end measurement bundle: tests, bench (+cpu baseline), rocprofv3 kernel stats, PMC traffic.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final gpurun_out/pmc
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || exit 1
cat gpurun_out/final/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final -o prof -- python bench.py --steps 35 --warmup 7 --no-cpu-baseline > gpurun_out/final/bench_under_rocprof.json 2> gpurun_out/final/prof_err.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc -o pmc_$c -- python bench.py --steps 14 --warmup 7 --no-cpu-baseline > gpurun_out/pmc/bench_$c.json 2> gpurun_out/pmc/err_$c.log || exit 1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc/pmc_%s_counter_collection.csv" % c)
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(f[0]) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmc/pmc_summary.json", "w"), indent=1)
PY
find gpurun_out -name '*.csv' -size +6M -delete
# per-kernel summary of the rocprofv3 run, MFMA utilisation, traffic per launch
python tools/pmc_to_traffic.py gpurun_out/pmc/pmc_summary.json gpurun_out/final/pmc_traffic.json > /dev/null
ls gpurun_out/final
